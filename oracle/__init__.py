"""CPU oracle for the ARFlow hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package restates, in plain PyTorch running on the CPU, the arithmetic of the
reference's cost-volume correlation, bilinear warp, forward-splat occlusion maps and
unsupervised photometric / smoothness losses (deu439/ARFlow: models/correlation_native.py,
utils/warp_utils.py, utils/uflow_utils.py, utils/uflow_resampler.py, losses/loss_blocks.py,
losses/uflow_loss.py, losses/flow_loss.py, losses/fullres_loss.py).  Every function cites the
reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- always as the *checker*, never as the thing measured or shipped.  Nothing under
``arflow_amd/`` imports this package; the product path has no CPU fallback and raises when the
HIP library is missing or a tensor is not on the GPU.

Parity pin: the oracle is pinned against the reference itself.  ``oracle/make_golden.py``
imports the reference's pure-PyTorch modules from ``/root/reference`` (possible only in the
build container; the reference never travels) and freezes inputs + reference outputs +
reference autograd gradients as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
every oracle function against those vectors, and ``tests/test_oracle_vs_reference.py`` re-runs
the comparison live whenever ``/root/reference`` is present.  The bilinear sampler, pooling and
resize arithmetic that the reference delegates to torch (``F.grid_sample``, ``scatter_add_``,
``AvgPool2d``, ``F.interpolate``; torch is unpinned in the reference's requirements.txt:5, the
version here is 2.10.0) is restated explicitly in ``oracle/ops.py`` and additionally checked
against the installed torch functions.
"""
