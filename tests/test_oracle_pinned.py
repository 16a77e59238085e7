"""CPU, build container only: the oracle is torch.equal to the imported reference.
Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest

from util import ROOT

REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')
def test_oracle_bit_equal_to_reference(tmp_path):
    code = r'''
import sys, types, os, tempfile
import numpy as np, torch
sys.dont_write_bytecode = True
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, 'tests', 'golden'))
import make_golden as mg
mg.import_reference()
from graph_neural_net_amd import synthetic
model = mg.build_reference_model(2, seed=5)
mg.perturb_(model, 55)
x1, x2 = synthetic.make_batch(77, 3, 17, 'ErdosRenyi', 0.3, 0.1)
worst = mg.check_oracle_bit_equal(model, x1, x2, 'live')
assert worst == 0.0, worst
print('PINNED')
''' % (ROOT, ROOT)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and 'PINNED' in out.stdout, out.stderr[-2000:]
