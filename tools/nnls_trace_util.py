#!/usr/bin/env python3
"""Lane-pass utilisation of one NNLS launch from an OPTIK_NNLS_TRACE dump: the solve passes the
problems needed against 16 x the passes their waves ran (a wave runs as many passes as its
slowest problem)."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 1] > 0]
summax = (a[:,3] & 0xFFFF).astype(np.int64)
sump = ((a[:,2] >> 40) & 0xFFFF).astype(np.int64)
npb = ((a[:,2] >> 56) & 0xFF).astype(np.int64)
print("waves", len(a), "problems", npb.sum(), "sum passes", sump.sum(), "mean passes/problem", sump.sum()/max(npb.sum(),1))
print("wave pass-slots (16 per pass)", 16*summax.sum(), "utilisation", sump.sum()/(16*summax.sum()))
