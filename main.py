#!/usr/bin/env python
"""`python main.py --config ... [--test]` — same command line as the reference's main.py; runs the
MI355X-native solvers in end-to-end-asr-pytorch_amd/."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == '__main__':
    importlib.import_module("end-to-end-asr-pytorch_amd.main").main()
