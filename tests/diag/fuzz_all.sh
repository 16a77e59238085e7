for s in 71 72 73 74; do echo "models wide $s"; python tests/diag/gpu_fuzz_models.py 80 $s wide 2>&1 | grep -E "FAIL|failed"; done
for s in 75 76; do echo "models $s"; python tests/diag/gpu_fuzz_models.py 80 $s 2>&1 | grep -E "FAIL|failed"; done
for s in 2 3 4; do echo "conv $s"; python tests/diag/gpu_fuzz_conv.py 300 $s 2>&1 | grep -E "FAIL|failed"; done
for s in 81 82 83; do echo "shapes $s"; python tests/diag/gpu_fuzz_shapes.py 60 $s 2>&1 | grep -E "FAIL|failed"; done
for s in 6 7; do echo "shapes big $s"; python tests/diag/gpu_fuzz_shapes.py 24 $s big 2>&1 | grep -E "FAIL|failed"; done
for s in 91 92 93; do echo "struct $s"; python tests/diag/gpu_fuzz_struct.py 60 $s 2>&1 | grep -E "FAIL|failures" | tail -3; done
for s in 11 12; do echo "surface $s"; python tests/diag/gpu_fuzz_surface.py 60 $s 2>&1 | grep -E "FAIL|failures" | tail -2; done
