"""Phase split of the similarity column kernel (MI355REC_SIM_PHASES=1) at ML-20M shape, binary and real-valued data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI355REC_SIM_PHASES"] = "1"
import numpy as np
from bench import load_urm, TOPK
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
urm = load_urm("ml20m")
real = urm.copy(); real.data = (1 + (np.arange(real.nnz) % 5)).astype(np.float32)
for name, X in (("binary (ds_add_u32 counts)", urm), ("integer ratings 1..5 (exact int32 sums)", real)):
    s = Compute_Similarity_MI355X(X, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
    s.compute_slabs()
    print("ML-20M shape, cosine, topK=100,", name, flush=True)
    s.compute_slabs()
    s.close()
