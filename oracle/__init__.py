"""CPU oracle for the multi-pass GAN hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy restatement of the arithmetic the reference
(maxwerhahn/Multi-pass-GAN) issues through TensorFlow 1.x on its hot path:
the 2D generator conv stacks, the legacy-TF resize ops, the axis zoom and the
volume <-> slice-batch marshalling.  Every function cites the reference
file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product package
(``multi-pass-gan_amd``) never does: its compute runs in hand-written HIP
kernels behind the C ABI of ``include/mpgan.h`` and fails loudly when that
library is missing.

Pinning status
--------------
* ``.uni`` codec, ``FluidDataLoader`` marshalling, ``TileCreator`` tiles and the
  ``scipy.ndimage.zoom`` axis interpolation are pinned by golden fixtures
  generated HERE by importing the reference's own TensorFlow-free tool
  modules (``tests/golden/make_golden.py``, fixtures in ``tests/golden``).
* The conv stacks themselves live behind ``import tensorflow`` in the
  reference (``tools_wscale/GAN.py:13``); TensorFlow 1.x is not installable
  in this image and the reference holds no test vectors for them, so for the
  conv / resize / batch-norm arithmetic this oracle is **parity unpinned**:
  it restates TF 1.x semantics from their published definition and is
  cross-checked against an independent PyTorch-CPU implementation
  (``oracle/torch_ref.py``) and authored known-answer tests.
* The training graphs (``oracle/train_ref.py``: 4x GAN step, temporal discriminator,
  ``tensorResample``, TF-flavoured Adam; ``oracle/train_ref8x.py``: progressive-growing nets,
  WGAN-GP) are float64 PyTorch-autograd restatements, **parity unpinned** for the same reason;
  their gradients are checked against central finite differences and ``tensorResample`` against a
  scalar loop (``tests/test_oracle.py``).
"""
