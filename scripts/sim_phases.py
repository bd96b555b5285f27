"""Phase split of the similarity column kernel (MI355REC_SIM_PHASES=1) at ML-20M shape, binary and integer-rating data, with the
threshold-first selection on and off; then the un-instrumented kernel time of each (best of 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_urm, TOPK
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
urm = load_urm(sys.argv[1] if len(sys.argv) > 1 else "ml20m")
real = urm.copy(); real.data = (1 + (np.arange(real.nnz) % 5)).astype(np.float32)
cases = (("binary (ds_add_u32 counts)", urm), ("integer ratings 1..5 (exact int32 sums)", real))
if os.environ.get("SIM_PHASES_BINARY_ONLY"):
    cases = cases[:1]
for name, X in cases:
    s = Compute_Similarity_MI355X(X, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
    s.compute_slabs()
    for fast in ("1", "0"):
        os.environ["MI355REC_SIM_FAST_TOPK"] = fast
        os.environ["MI355REC_SIM_PHASES"] = "1"
        print("%s shape, cosine, topK=100, %s, threshold-first selection %s" % (sys.argv[1] if len(sys.argv) > 1 else "ml20m", name, "on" if fast == "1" else "OFF"), flush=True)
        sys.stderr.flush()
        s.compute_slabs()
        del os.environ["MI355REC_SIM_PHASES"]
        best = min((s.compute_slabs(), s.stats()["kernel_ms"])[1] for _ in range(5))
        print("    un-instrumented kernel: %.3f ms (best of 5); selection_info %s" % (best, s.selection_info(),), flush=True)
    del os.environ["MI355REC_SIM_FAST_TOPK"]
    s.close()
