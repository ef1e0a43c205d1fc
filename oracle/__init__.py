"""CPU oracle for the FMC denoising hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 restatement of the reference algorithm
(FudanCVL/SynFMC, `/root/reference`, snapshot 2025-08-24).  It exists so that
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
can check / time the hand-written HIP path against something that follows the
reference line by line.  Nothing under `synfmc_amd/` may import it: the product
path fails loudly when `libfmc_hip.so` is missing, it never falls back here.

Pinning status:

* `oracle.conditioning` (Pluecker rays, OMC rasteriser, Gaussian circle mask,
  relative-pose math) and `oracle.fmc_modules.Adapter` are pinned against the
  reference itself: `tests/golden/make_golden.py` imports the reference's
  `fmc.data.dataset.ray_condition`, `fmc.util.get_traj_features_v2`,
  `fmc.adapter.Adapter` and `fmc.data.utils` in the build container and the
  resulting vectors are committed under `tests/golden/`.
* Everything that the reference delegates to `diffusers==0.24.0`
  (`environment.yaml:13`; not installed, no network) is restated from the
  published algorithm in `oracle.diffusers_restated` -- **parity unpinned** for
  those primitives (the reference has no tests or golden vectors for them).
  They are cross-checked against torch built-ins in `tests/test_oracle_*.py`.
* The reference-owned orchestration that *uses* those primitives
  (`fmc/models/*.py`, `fmc/modified_modules.py`) is additionally pinned by
  running the reference's own source over `oracle.diffusers_restated` exposed
  as a stand-in `diffusers` namespace (`tests/golden/make_golden_g5.py`, vectors `g5_*`);
  that pins the reference-owned arithmetic only, never diffusers' own.
"""
