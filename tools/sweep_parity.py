"""The random-configuration parity sweep of tests/test_gpu_parity.py over more seeds than the suite runs
(40..199), one process; prints the configurations that fail."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_parity import test_random_configurations_against_oracle as t
dev = torch.device('cuda')
bad = []
for s in range(40, 200):
    try:
        t.__wrapped__(dev, s) if hasattr(t, '__wrapped__') else t(dev, s)
    except Exception as e:
        bad.append((s, str(e)[:200]))
    if s % 20 == 0: print('seed', s, 'failures so far', len(bad), flush=True)
print('done; failures:', bad)
