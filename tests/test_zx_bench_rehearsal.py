"""bench.py's N > 1 code path, rehearsed on ONE card in two real processes (the N > 1 run itself belongs to the driver).

Collected near the end on purpose (only tests/test_zz_no_alloc.py comes later): the rehearsal starts torch.distributed.run
with two ranks that share the GPU and talk over gloo -- more moving parts than any other test here -- and the round's
end runs `pytest -x`.  A rank that stalls dumps its stacks and exits (HVRT_BENCH_WATCHDOG) instead of holding the run.
"""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_multi_rank_code_path_rehearsed_on_one_gpu(scaling):
    """bench.py as it runs for N > 1 (torch.distributed.run, one rank per GPU; the launcher picks its own rendezvous port here,
    the driver passes --master-addr / --master-port -- bench.py reads RANK / WORLD_SIZE / MASTER_* either way), rehearsed on this box's one
    GPU: HVRT_BENCH_REHEARSE=1 puts both ranks on device 0 and stages the collectives through gloo.  What this pins is
    that the N > 1 code of bench.py -- scene replication, 8-bit batch pipeline (weak) / one frame over all ranks (strong),
    max-over-ranks timing, the JSON line -- runs end to end; the numbers of such a run mean nothing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HVRT_BENCH_REHEARSE="1", HVRT_BENCH_WATCHDOG="150", HVRT_TRACE_BUILD="1")      # a rank that stalls dumps its stacks and exits
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
                        "--nproc-per-node", "2", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "small", "--scaling", scaling], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-6000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl"]["world_size"] == 2 and d["rccl"]["backend"] == "gloo" and d["scaling"] == scaling
    assert d["value"] > 0 and d["unit"] == "Mrays/s" and d["config"]["workload"] == "small"
    assert d["config"]["frames_per_step"] == (2 if scaling == "weak" else 1) and "rgb8" in d["rccl"]["exchange"]
    # the line of an N > 1 run carries a roofline object too: rank 0's last launch, its share of the step's rays
    rf = d["roofline"]
    assert rf["kernel"] == "k_render" and rf["kernel_ms"] > 0 and "rank 0" in rf["launch"]
