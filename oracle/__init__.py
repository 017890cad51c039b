"""CPU oracle for the MGDT-YOLO detection hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain CPU restatement (torch-CPU / numpy, fp32) of the reference's algorithm for
the path named in BASELINE.json.  It exists to CHECK the HIP product path; it is never the thing
shipped or measured.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it.  The product package (`mgdt_yolo_amd`) never imports it and fails loudly when
its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * every function here except the two below is pinned by fixtures under tests/golden/ that were
    generated in the build container by importing the reference's own Python modules
    (tests/golden/gen_golden.py, recipe in tests/golden/ref_import.py);
  * `nms.greedy_nms` restates `torchvision.ops.nms` (call site yolo/utils/ops.py:249); torchvision
    is not vendored in the reference, not pinned to a version and not installed here
    -> PARITY UNPINNED at that call; the stages of `non_max_suppression` around it ARE pinned;
  * TOODHead / DCNv2 (mmcv, unpinned, absent) is not restated in this round.

Each function cites the reference file:line it follows (paths relative to /root/reference).
"""
