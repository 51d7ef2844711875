import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
import numpy as np, torch
from humanoid.envs import task_registry
from humanoid.utils import get_args
from humanoid.algo import PPO
PPO.precision = "bf16"
real = torch.randperm
def run(mode, iters):
    os.environ["HGYM_GRAPH"] = mode
    torch.manual_seed(1234); np.random.seed(1234)
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "256", "--seed", "77"])
    env, _ = task_registry.make_env(name=args.task, args=args)
    r, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    cnt = [0]
    def fixed(n, *a, **k):
        g = torch.Generator().manual_seed(cnt[0]); cnt[0] += 1
        return real(n, generator=g).to(k.get("device", "cpu"))
    torch.randperm = fixed
    r.env.episode_length_buf = torch.arange(256, device="cuda") * 9
    snaps = []
    for it in range(iters):
        r.learn(num_learning_iterations=1, init_at_random_ep_len=False)
        torch.cuda.synchronize()
        st = r.alg.storage
        snaps.append(dict(params=r.alg.net.params.clone(), obs=st._obs_all.clone(), rew=st.rewards.clone(), act=st.actions.clone(),
                          val=st.values.clone(), adv=st.advantages.clone(), dones=st.dones.clone()))
    torch.randperm = real
    return snaps
a = run("0", 3); b = run("0", 3); c = run("1", 3)
for name, x, y in (("eager-vs-eager", a, b), ("eager-vs-graph", a, c)):
    for it in range(3):
        for k in x[it]:
            if not torch.equal(x[it][k], y[it][k]):
                d = (x[it][k].float() - y[it][k].float()).abs()
                print(name, "iter", it, k, "DIFF max %.3e count %d of %d" % (float(d.max()), int((d > 0).sum()), d.numel()))
print("done")
