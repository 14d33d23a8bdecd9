import sys, numpy as np
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import indelope_amd, oracle
from indelope_amd import synth
from test_gpu_round6 import _many_contig_variant
hip = indelope_amd.api(); hip.init(0)
raw, _ = synth.generate(600, n_reads=(64, 64), err_rate=1e-3, config_id=62)
clean, dirty = raw.with_trim_bounds(), _many_contig_variant(raw, 7).with_trim_bounds()
hip.debug_set()
for k in range(8):
    bt = dirty if k % 2 else clean
    h = hip.batch_upload(bt)
    hip.batch_run(h); hip.batch_sync(h)
    p = hip.batch_profile(h)
    print(k, "skipped", p[21], "back", p[23], "big", p[28], "b", p[29], "c", p[30], "reruns", p[31], "retry123", p[24:27])
    hip.batch_free(h)
