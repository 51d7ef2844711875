"""Does the 256 MiB Infinity Cache carry a minibatch chunk from the forward+backward kernel to the weight-gradient kernel?
Runs the default update (4 minibatches of 61 440 rows: 0.62 GB of activations per minibatch) and the same with 16 / 32
minibatches (155 / 78 MB each) and prints per-kernel time per ROW -- if the smaller working sets are served on-die,
dw_kernel_rs gets cheaper per row."""
import io, os, sys, contextlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "humanoid-gym_amd")]
import torch
from humanoid.envs import task_registry
from humanoid.utils import get_args
from hgym import _lib as L

os.environ["HGYM_GRAPH"] = "0"
for mb in [int(x) for x in (sys.argv[1:] or ["4", "16", "32"])]:
    a = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "4096"])
    with contextlib.redirect_stdout(io.StringIO()):
        env, _ = task_registry.make_env(name=a.task, args=a)
        _, tc = task_registry.get_cfgs(a.task)
        tc.algorithm.num_mini_batches = mb
        runner, _ = task_registry.make_alg_runner(env=env, name=a.task, args=a, train_cfg=tc, log_root=None)
        runner.learn(num_learning_iterations=2, init_at_random_ep_len=True)
        torch.cuda.synchronize()
        L.lib.hgym_prof_enable(1)
        runner.learn(num_learning_iterations=2, init_at_random_ep_len=False)
        torch.cuda.synchronize()
    rows = 4096 * 60 // mb
    out = []
    for cid, nm in [(L.PROF_MLP_FWD, "fb"), (L.PROF_DW, "dw"), (L.PROF_REDUCE, "reduce"), (L.PROF_APPLY, "apply")]:
        n, ms, _ = L.prof_summary(cid)
        if n:
            out.append("%s %7.1f us (%6.2f ns/row, n=%d)" % (nm, ms / n * 1e3, ms / n * 1e6 / rows, n))
    print("minibatches=%2d rows=%6d  update %.2f ms | %s" % (mb, rows, runner.last_learn_time * 1e3, " | ".join(out)), flush=True)
    L.lib.hgym_prof_enable(0)
    del runner, env
    torch.cuda.empty_cache()
