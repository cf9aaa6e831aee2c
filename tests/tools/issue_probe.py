import sys, ctypes as C
sys.path.insert(0, "drl-on-robot-arm_amd")
import torch
from armenv import _lib
lib = _lib.load()
for rep in range(2):
    for prec, w in ((64, 1), (64, 2), (64, 4), (32, 1), (32, 2)):
        v = C.c_double()
        rc = lib.armenv_probe_issue_rate(0, prec, w, C.byref(v))
        print(rep, prec, w, rc, "%.3f ns = %.2f cycles at 2.4 GHz" % (v.value, v.value * 2.4))
