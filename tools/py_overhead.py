import sys, os, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import numpy as np, torch, metrpo_amd
from metrpo_amd import synthetic
cfg = synthetic.CONFIGS['C1']
env, K, B, H = cfg['env'], cfg['K'], cfg['B'], cfg['H']
eng = metrpo_amd.Engine(env, K, cfg['dyn_hidden'], cfg['pol_hidden'])
Ws, bs, norm = synthetic.make_dynamics(env, K, cfg['dyn_hidden'], seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
policy = metrpo_amd.GaussianMLPPolicy(eng, init_std=1.0, seed=0)
init = metrpo_amd.InitStatePool(synthetic.make_pool(env), eng.na)
nne = metrpo_amd.NeuralNetEnv(env=init, inner_env=None, cost_np=env, dynamics_in=None, dynamics_outs=eng, sam_mode='step_rand')
algo = metrpo_amd.TRPO(env=nne, policy=policy, baseline=metrpo_amd.LinearFeatureBaseline(), batch_size=B * H, max_path_length=H, discount=1.0,
                       step_size=0.01, sampler_args=dict(n_envs=B), seed=0)
algo.defer_baseline_fit = True; algo.reuse_trajectory_buffers = True
def step(j):
    algo.start_worker(); paths = algo.obtain_samples(j); s = algo.process_samples(j, paths); algo.optimize_policy(j, s)
for j in range(5): step(j)
torch.cuda.synchronize()
# CPU time of the launch path between the end of optimize_policy and the rollout launch
ts = []
for j in range(5, 25):
    t0 = time.perf_counter(); algo.start_worker(); t1 = time.perf_counter(); paths = algo.obtain_samples(j); t2 = time.perf_counter()
    s = algo.process_samples(j, paths); t3 = time.perf_counter(); algo.optimize_policy(j, s); t4 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
a = np.array(ts) * 1e6
print('CPU us per call: start_worker %.0f  obtain_samples %.0f  process_samples %.0f  optimize_policy %.0f' % tuple(a.mean(0)))
pr = cProfile.Profile(); pr.enable()
for j in range(25, 45): step(j)
pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(18)
