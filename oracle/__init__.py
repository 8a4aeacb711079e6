"""CPU oracle package — TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product path (end-to-end-asr-pytorch_amd/) never does."""
