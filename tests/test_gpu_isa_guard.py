"""The mechanical guards of tests/test_isa_guard.py once more inside `-m gpu`: they need no GPU (llvm-objdump on the built
libarmenv.so), but the round driver runs only the GPU suite on the MI355X box, and it is THAT box's copy of the library
whose code object has to honour the inline-asm patterns (AGPR action prefetch, the f16x3 k-loop's DMA ring, no scratch in
the step kernels).  Skipped where the ROCm LLVM tools are absent."""
import pytest

from test_isa_guard import *          # noqa: F401,F403  (the `kernels` fixture and every test_* function)

pytestmark = pytest.mark.gpu
