"""Solver defaults (reference: src/option.py:2-11 — same keys and values)."""
default_hparas = {
    'GRAD_CLIP': 5.0,          # gradient-norm clip threshold
    'PROGRESS_STEP': 100,      # stdout / log refresh period (steps)
    'DEV_STEP_RATIO': 1.2,     # validation decodes ratio * longest transcript steps
    'DEV_N_EXAMPLE': 4,        # number of hypotheses logged per validation
    'TB_FLUSH_FREQ': 180       # log flush period (secs)
}
