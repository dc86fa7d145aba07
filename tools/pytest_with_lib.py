"""Run the GPU parity tests against another build of libdpfhe_hip.so (tool for same-box A/B work):
    python tools/pytest_with_lib.py deeppowers_amd/csrc/build/var_x.so tests/test_gpu_parity.py -m gpu -q -k ntt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
_cabi.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
