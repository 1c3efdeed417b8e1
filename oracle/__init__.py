"""CPU oracle for the TRIBE trimodal-encode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`algonauts-2025_amd/`) imports this; only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may.  It is the checker, never the
thing measured or shipped.

Contents
--------
* `tribe_ref`   -- fp32 PyTorch-CPU restatement of the reference arithmetic,
                   function by function, each citing the reference file:line.
* `xt_encoder`  -- restatement of the third-party `x_transformers.Encoder`
                   (absent from /root/reference and from this image) as an
                   `nn.Module` with the library's state_dict key names.

Pinning status (see DESIGN.md "Oracle")
---------------------------------------
* Everything that lives in the reference's own files (SubjectLayers,
  aggregate_features, FmriEncoder.forward glue, PearsonLoss, InfoNCE, the
  `_run_step` flatten order, `_aggregate_layers`) is PINNED by golden vectors
  produced by executing those very files (tests/golden/make_golden.py).
* Per-voxel Pearson is PINNED against `scipy.stats.pearsonr`, which is what
  the reference itself calls (algonauts2025/main.py:474-477).
* The encoder internals (`x_transformers`, not vendored, not installed) and
  the `torchmetrics.PearsonCorrCoef` streaming update are PARITY UNPINNED:
  the restatement here defines them.
"""
