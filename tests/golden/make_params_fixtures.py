#!/usr/bin/env python3
"""Key sets of the reference's run configurations, as test fixtures: reads /root/reference/params/params-<env>.json (build container only), drops the
free-text `_comment*` entries, and writes tests/golden/params_<env>.json.  tests/test_params.py loads these through metrpo_amd.shapes_from_params."""
import json
import os

REF = '/root/reference/params'
HERE = os.path.dirname(os.path.abspath(__file__))


def strip(o):
    if isinstance(o, dict):
        return {k: strip(v) for k, v in o.items() if not k.startswith('_comment')}
    return o


if __name__ == '__main__':
    for fn in sorted(os.listdir(REF)):
        if fn.startswith('params-') and fn.endswith('.json'):
            name = fn[len('params-'):-len('.json')].replace('-', '_')
            with open(os.path.join(REF, fn)) as f:
                d = strip(json.load(f))
            with open(os.path.join(HERE, 'params_%s.json' % name), 'w') as f:
                json.dump(d, f, indent=1, sort_keys=True)
            print(name, sorted(d))
