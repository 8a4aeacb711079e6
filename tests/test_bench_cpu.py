"""CPU: bench.py's workload definitions and algorithmic-work formulas against the figures of
SURVEY.md §8(d) (the numbers `roofline` / `encoder_fwd` are computed from), and the CLI contract."""
import importlib

import pytest
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_workloads_and_algorithmic_work_match_survey_table():
    bench = importlib.import_module("bench")
    w2, w3 = bench.WORKLOADS["cfg2"], bench.WORKLOADS["cfg3"]
    assert (w2["B"], w2["T"], w2["D"], w2["V"], w2["L"]) == (32, 1000, 80, 5000, 64)
    assert (w3["B"], w3["T"], w3["D"], w3["V"], w3["L"]) == (32, 1600, 80, 5000, 64)
    a2, a3 = bench.encoder_algorithmic_work(w2), bench.encoder_algorithmic_work(w3)
    # cfg2 encoder 0.390 GB / 0.289 T ih / 0.201 T hh / 1500 steps;
    #                  cfg3 encoder 2.077 GB / 3.074 T ih / 1.611 T hh / 3000 steps
    assert abs(a2["bytes"] / 1e9 - 0.390) < 0.001 and a2["steps"] == 1500
    assert abs(a2["flops_ih"] / 1e12 - 0.289) < 0.001 and abs(a2["flops_hh"] / 1e12 - 0.201) < 0.001
    assert abs(a3["bytes"] / 1e9 - 2.077) < 0.001 and a3["steps"] == 3000
    assert abs(a3["flops_ih"] / 1e12 - 3.074) < 0.001 and abs(a3["flops_hh"] / 1e12 - 1.611) < 0.001
    assert bench.F32_MFMA_PEAK_TFLOPS == 157.3


def test_bench_cli_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"',
                '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"',
                '"roofline"', '"cpu_baseline"', '"roofline_hbm"', '"rccl"'):
        assert key in src, key


def _run_bench(argv, env_extra=None, timeout=240):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                       env=env, timeout=timeout)
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    return r, [json.loads(x) for x in lines]


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it (no WORLD_SIZE) must itself start two ranks - the flag
    used to be parsed and ignored.  --dry-spawn runs the launch path on CPU (gloo): both ranks rendezvous on
    127.0.0.1, all-reduce their rank numbers (0 + 1) and rank 0 alone prints ONE line."""
    r, lines = _run_bench(["--gpus", "2", "--dry-spawn"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["rccl"]["world_size"] == 2
    assert out["rccl"]["rank_sum_allreduce"] == out["rccl"]["rank_sum_expected"] == 1.0
    assert sorted(x["rank"] for x in out["ranks"]) == [0, 1]
    assert sorted(x["local_rank"] for x in out["ranks"]) == [0, 1]        # rank r is bound to GPU r
    assert len({x["pid"] for x in out["ranks"]}) == 2                      # two processes


def test_bench_refuses_a_world_that_differs_from_gpus_flag():
    """under an external launcher (the driver's torch.distributed.run form) WORLD_SIZE must equal --gpus: a line
    claiming N GPUs from another number of ranks is never printed"""
    r, lines = _run_bench(["--gpus", "4", "--dry-spawn"],
                          {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                           "MASTER_PORT": "29999"})
    assert r.returncode != 0 and not lines
    assert "--gpus 4" in r.stderr


def test_stamped_hbm_traffic_of_the_default_workload_describes_these_kernel_sources():
    """bench.py reports `roofline.traffic` from the newest profiles/rNN_hbm_traffic_<workload>.json only while the
    summary's kernel-source digest equals the digest of the tracked kernel sources (a stale summary gives null): the
    committed evidence must have been collected on the committed kernels."""
    import glob
    import json
    bench = importlib.import_module("bench")
    stale = []
    for workload in ("cfg3", "shipped"):
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_%s.json" % workload)), reverse=True)
        cands = [c for c in cands if "f32mfma" not in os.path.basename(c)]
        assert cands, workload
        tj = json.load(open(cands[0]))
        assert any(k.startswith("gemm_bf16x6") for k in tj["kernels"])
        if tj["kernel_source_digest"] != bench.kernel_source_digest():
            # a kernel source changed after the counters were collected: bench.py then reports `traffic: null` (checked
            # below) until tools/pmc_hbm.sh has been re-run on a gfx950 box - not something a CPU-only check-out can do
            stale.append(os.path.basename(cands[0]))
    path, tj = bench.newest_traffic_summary("cfg3", bench.kernel_source_digest())
    assert path and (tj is None) == ("cfg3" in " ".join(stale))      # a stale summary is dropped, a current one is used
    if stale:
        pytest.skip("stale HBM-traffic summaries (re-run tools/pmc_hbm.sh on the GPU box): %s" % ", ".join(stale))
