"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's mPLUG-Video pre-train hot path used as
the parity checker for the HIP product path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import anything from here; the product package
(youku-mplug_amd/) never does and fails loudly when its HIP library is missing.

Contents
  shims/        stand-ins for un-installable third-party packages (megatron_util, timm,
                addict, utils.File) so the reference's own modules import on CPU.
  ref_loader.py imports the reference modules UNMODIFIED from /root/reference (this
                container only) and builds DistributedGPT3_Pretrain on CPU.
  restate.py    independent functional PyTorch restatement of the same algorithm (travels
                to the GPU box; validated here against ref_loader by tests and gen_golden).
  weights.py    seeded weight/input generators shared by all of the above.
  gen_golden.py writes tests/golden/*.pt from the real reference modules.

Parity pinning: the reference ships NO tests/goldens, so the oracle is
pinned against outputs of the reference's own modules run here (tests/golden/), with the
megatron_util arithmetic restated from the published Megatron-LM semantics -- that one
boundary is "parity unpinned" (see shims/megatron_util/__init__.py).
"""
