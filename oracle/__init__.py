"""CPU oracle for the W-HMR hot path.  TEST INFRASTRUCTURE ONLY.

Pure-PyTorch fp32 restatement of the reference algorithm (yw0208/W-HMR) for the
path named in BASELINE.json:north_star.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this package; the shipped
product (``w-hmr_amd/``) never does and fails loudly without its HIP library.

Parity status (see DESIGN.md "Oracle"):
  * W-HMR-authored arithmetic (ViT, deconv pyramid, MAF sampler, regressor loop,
    geometry, Tz head, global-orient head, output dict) is PINNED: checked in the
    build container against the imported reference (tests/golden/make_golden.py)
    and against the committed fixtures tests/golden/*.npz.
  * Arithmetic living in un-vendored third parties (smplx==0.1.28 LBS + joint
    selector, pare==0.1 SMPL wrapper / softargmax1d / batch_euler2matrix /
    resnet50, timm==0.4.9 Block) is restated from the published algorithms and
    the in-tree spec (models/smpl_webuser/lbs.py) -- "parity unpinned" for those
    pieces; they are pinned by analytic known-answer tests instead.
"""
