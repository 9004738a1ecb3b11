"""W-HMR forward path on MI355X (gfx950): hand-written HIP kernels behind the reference's module API.

Import as ``whmr_amd`` (repo-root alias of this directory).  Sub-packages mirror the reference layout for the
hot path only: ``models`` (whmr_net / WHMR, pose_vit, maf_extractor, smpl), ``utils.geometry``, ``core.cfgs``.
The compute lives in ``csrc/*.hip`` -> ``libwhmr_hip.so`` (C ABI: include/whmr_hip.h), bound by ``_lib``.
"""
__version__ = '0.1.0'
