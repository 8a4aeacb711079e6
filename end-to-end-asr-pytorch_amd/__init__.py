"""MI355X-native LAS/CTC training + decode hot path (drop-in behind the reference's
main.py / src.asr / src.solver surface).  The directory name carries hyphens, so import it with

    import importlib; pkg = importlib.import_module("end-to-end-asr-pytorch_amd")

Sub-modules: ``ops`` (autograd Functions over the libasrk C ABI), ``src`` (mirror of the
reference's ``src`` package: asr, module, ctc, audio, decode, optim, solver, ...), ``bin``
(Solvers), ``parallel`` (data-parallel engine over RCCL).
"""
__version__ = "0.1.0"
