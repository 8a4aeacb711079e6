"""CPU: bench.py's workload definitions and algorithmic-work formulas against the figures of
SURVEY.md §8(d) (the numbers `roofline` / `encoder_fwd` are computed from), and the CLI contract."""
import importlib
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
                '"roofline"', '"cpu_baseline"'):
        assert key in src, key
