"""Dataset locations -- same contract as the reference's lvae/paths.py:8-32 (name -> Path under a
`datasets` root three levels above the package; override with the LVAE_DATASETS environment variable)."""
import os
from pathlib import Path

_root = Path(os.environ.get('LVAE_DATASETS', (Path(__file__).parent / '../../../datasets'))).resolve()

known_datasets = {
    'kodak': _root / 'kodak',
    'clic2022-test': _root / 'clic/test-2022',
    'tecnick-rgb-1200': _root / 'tecnick/TESTIMAGES/RGB/RGB_OR_1200x1200',
    'coco-train2017': _root / 'coco/train2017',
    'coco-val2017': _root / 'coco/val2017',
    'imagenet-train': _root / 'imagenet/train',
    'imagenet-val': _root / 'imagenet/val',
    'vimeo-90k': _root / 'vimeo-90k/sequences',
    'uvg-1080p': _root / 'video/uvg/1080p-frames',
}
