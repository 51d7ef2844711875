"""GPU sanity run: N learning iterations of the drop-in stack at the BASELINE size with per-iteration losses read back
(sync path), printing every 10th: value loss falls, learning rate adapts, nothing goes non-finite."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "humanoid-gym_amd"))
import torch
from humanoid.envs import task_registry
from humanoid.utils import get_args
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
every = int(os.environ.get("HGYM_SANITY_EVERY", "10"))             # HGYM_PRECISION=f32: the reference's own arithmetic, same seed
task = sys.argv[2] if len(sys.argv) > 2 else "humanoid_ppo"       # humanoid_dwl_ppo: + the denoising head's MSE
a = get_args(["--task=" + task, "--headless", "--num_envs", "4096"])
env, _ = task_registry.make_env(name=a.task, args=a)
runner, _ = task_registry.make_alg_runner(env=env, name=a.task, args=a, log_root=None)
alg = runner.alg
for it in range(iters):
    os.environ["HGYM_ASYNC"] = "0"
    runner.learn(num_learning_iterations=1, init_at_random_ep_len=(it == 0))
    o = alg.net.opt_state.cpu()
    if it % every == 0 or it == iters - 1:
        st = alg.storage
        n = max(float(o[7]), 1.0)
        print("it %4d  value_loss %.5f  surrogate %+.5f  kl %.5f  lr %.2e  |grad| %.3f  mean_rew/step %.4f  std %.3f  ep_len %.1f  finite %s%s" % (
            it, float(o[4]) / n, float(o[3]) / n, float(o[2]) / n, float(o[0]), float(o[6]), float(st.rewards.mean()),
            float(alg.actor_critic.std.detach().mean()), float(env.episode_length_buf.float().mean()), bool(torch.isfinite(alg.net.params).all()),
            ("  denoise_mse %.5f" % (float(o[10]) / n)) if "dwl" in task else ""), flush=True)
