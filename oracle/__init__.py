"""CPU oracle for the MO-VAE training hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a plain PyTorch-CPU / numpy restatement of the reference's algorithm for the
per-step training path.  It exists only to *check* the HIP path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
    may import it; nothing under ``mo-vae_amd/`` does, and the product path raises if its HIP
    library is missing instead of falling back to anything in here;
  * it is pinned against golden vectors produced by importing the reference's own in-tree code
    (``tests/golden/generate_golden.py``): models, losses, MGDA and Aligned-MTL weightings;
  * the pieces whose arithmetic lives in the un-vendored third-party ``torchjd`` (floating
    ``git@main``; API of the 0.7 era) and ``quadprog==0.1.13`` / ``qpsolvers==4.8.1`` --
    ``autojac.mtl_backward`` / ``backward`` and ``UPGrad`` -- are restated from their published
    algorithm and pinned only by the docstring known-answer vectors the reference carries
    (utils/torchmoo/nupgrad.py:58-62) plus an independent scipy cross-check: **parity unpinned**
    beyond those.

Every function cites the reference ``file:line`` it follows (paths relative to the reference
root).
"""
