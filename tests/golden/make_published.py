"""Regenerates tests/golden/published/published_rd.json from the reference's own result files (build container only: reads
/root/reference/results/*).  The fixture holds numbers only -- lambdas, bpp, PSNR per (model, test set)."""
import json
import os

REF = '/root/reference/results'
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {'_about': 'Published rate-distortion points of the reference (duanzhiihao/lossy-vae, results/<set>/<set>-<model>.json): numbers only, '
                     're-keyed as cases[model][test-set] = {lambdas, bpp, psnr}.  Data fixture of scripts/accept-published.py; generated in the '
                     'build container by tests/golden/make_published.py.', 'cases': {}}
    for model in ('qarv_base', 'qres34m'):
        for ds in ('kodak', 'clic2022-test', 'tecnick-rgb-1200'):
            rel = f'{ds}/{ds}-{model}.json'
            j = json.load(open(os.path.join(REF, rel)))
            assert j['name'] == model and j['test-set'] == ds
            out['cases'].setdefault(model, {})[ds] = {'source': 'results/' + rel, 'lambdas': [float(x) for x in j['lambdas']],
                                                      'bpp': j['results']['bpp'], 'psnr': j['results']['psnr']}
    with open(os.path.join(HERE, 'published', 'published_rd.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
