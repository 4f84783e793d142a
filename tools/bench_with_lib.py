"""bench.py against another build of the native library (tools/build_exp.sh):  LVAE_LIB=_bin/<name>/liblvae_hip.so python tools/bench_with_lib.py [bench args]"""
import os, runpy, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
from lvae import _native
if os.environ.get('LVAE_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['LVAE_LIB'])
sys.argv = [os.path.join(REPO, 'bench.py')] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
