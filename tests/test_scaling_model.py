"""tools/scaling_model.py (verdict r5 item 2a): the model over the committed single-GPU measurements reproduces the committed curves and keeps its invariants."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_scaling_model_from_committed_measurements():
    import scaling_model as S
    meas = json.load(open(os.path.join(ROOT, 'profiles', 'r06_scaling_measurements.json')))
    committed = {m['config']: m for m in json.load(open(os.path.join(ROOT, 'profiles', 'r06_scaling_model.json')))['models']}
    for name, rows in meas.items():
        mo = S.model(name, rows)
        cfg = S.cfg_of(name)
        assert mo['per_gpu_share'] == cfg['B'] // cfg['gpus']
        for kind in ('weak', 'strong'):
            pts = [mo[kind][str(n)] for n in S.NS if str(n) in mo[kind]]
            vals = [p['value_env_steps_per_s'] for p in pts]
            assert all(b > a for a, b in zip(vals, vals[1:])), (name, kind, vals)           # more GPUs never lower the predicted throughput ...
            assert all(p['efficiency_vs_first_point'] <= 1.0 + 1e-9 for p in pts)                # ... and never scale super-linearly
            assert all(abs(sum(p['phases_ms'].values()) - p['ms_per_step']) < 1e-6 * p['ms_per_step'] for p in pts)
            for n in S.NS:
                if str(n) in mo[kind]:
                    assert abs(mo[kind][str(n)]['ms_per_step'] - committed[name][kind][str(n)]['ms_per_step']) < 1e-9
        # weak scaling: the per-GPU share is timed, only the exchanges are added; strong: B / N envs per rank
        assert [mo['strong'][str(n)]['B_per_gpu'] for n in S.NS] == [cfg['B'] // n for n in S.NS]
        assert mo['weak']['1']['phases_ms']['exchanges'] == 0.0 and mo['weak']['8']['phases_ms']['exchanges'] > 0.0
    # the Amdahl statement of DESIGN 7: C1's rollout does not shrink below one tile per CU
    c1 = {m['B']: m for m in meas['C1']}
    assert c1[625]['ms_rollout'] > 0.75 * c1[2500]['ms_rollout']
