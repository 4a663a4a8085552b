"""CPU oracle for the distillation-step hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 *restatement* (functional style, parameters passed as
``{state_dict_name: tensor}`` dicts) of the arithmetic of the reference's hot path
(SURVEY.md section 8a).  It is the checker the HIP kernels are compared against.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
The product package ``dinov2_distillation_amd`` never imports it and has no CPU fallback.

Parity pinning:
  * ``scalekd_oracle`` / ``step_oracle`` / ``convnext_oracle`` are PINNED: they are checked
    against golden vectors produced by importing the reference's own
    ``losses/scalekd.py``, ``train/distillation_module.py`` and ``models/backbones/convnext.py``
    (``tests/golden/make_golden.py`` is the generating script; fixtures in ``tests/golden/*.npz``).
  * ``vit_oracle`` (DINOv2 teacher) is **parity unpinned**: the teacher's arithmetic lives in the
    external ``facebookresearch/dinov2`` hub repository (unpinned, default branch), which is not under
    /root/reference and could not be fetched (denied - see DESIGN.md).  It restates the published
    architecture; the reference's call sites are ``models/backbones/dinov2.py:20,32,40`` and
    ``train/distillation_module.py:169-177``.
"""
