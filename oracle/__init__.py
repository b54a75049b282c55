"""oracle/ -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker.
The product package (``interactive_deep_colorization_amd``) never imports it
and has no CPU fallback: it fails loudly when the HIP library is missing.

Reference path restated here (all paths relative to the upstream repo):

* ``models/pytorch/model.py:134-175``  ``SIGGRAPHGenerator.forward``
* ``data/colorize_image.py:79-96,249-268`` ``ColorizeImageTorch.net_forward``
* ``data/colorize_image.py:20-36`` skimage Lab <-> RGB helpers
* ``DemoInteractiveColorization.ipynb:131-139`` ``put_point``

Pinning status
--------------
* ``siggraph_torch`` (the oracle proper): PINNED.  ``oracle/make_golden.py``
  imports the untouched reference module from ``/root/reference`` in the
  authoring container, loads the same numpy-seeded weights and checks the
  restatement against it (max-abs diff 0.0 in fp32), then writes the golden
  vectors under ``tests/golden/``.  The reference ships no tests, golden
  vectors or trained weights of its own, so these
  reference-generated fixtures are the only pins that exist.
* ``siggraph_numpy``: independent float64 restatement (im2col + matmul) used to
  cross-check the torch restatement and to measure the fp32 noise floor.
* ``colorspace`` (skimage restatement) and the Caffe-only branches inside
  ``siggraph_torch`` (``pred313_head``: 313-bin distribution + soft-decode,
  ``global_branch``: Global-Hints fusion; restatements of the prototxt):
  PARITY UNPINNED -- neither skimage nor Caffe can be installed here and the
  reference ships no weights or outputs for them; they follow the published
  formulas / the prototxt line by line (fixtures:
  ``oracle/make_golden_caffe_branches.py``).
* ``session`` (hint rasterisation = ``UIControl.get_input`` + ``rgb2lab``; colour suggestions =
  ``get_ab_reccs``): the rasteriser is PARITY UNPINNED (cv2 absent; ``cv2.rectangle`` semantics restated) and checked
  against the notebook's ``put_point`` run here; the suggestion step of the reference is stochastic (numpy global RNG +
  sklearn's randomly seeded KMeans), so ``session.get_ab_reccs_reference`` restates it AS WRITTEN and
  ``tests/test_session_cpu.py`` checks the deterministic form the device implements against it statistically.
"""
