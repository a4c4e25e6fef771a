"""CPU oracle for the Exposure filter-stack hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``exposure_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU path.

PARITY UNPINNED: the reference (``/root/reference``, TF-1 graph code) ships no
tests, golden vectors or fixtures, and cannot be imported in this image (no
tensorflow / cv2; ``util.py:658`` is a SyntaxError on Python >= 3.7).  The
restatement is therefore pinned by (i) closed-form identities of the reference
code, (ii) hand-computed points, (iii) float64 finite differences, and (iv)
agreement of three independently written restatements (``filters_np`` with
hand-derived backward, ``filters_torch`` with autograd, and since round 2
``filters_c.c``: plain C following the reference's formulas literally, also
the CPU baseline of ``bench.py``), and (v) for the two
TensorFlow image ops the reference calls but does not contain
(``tf.image.rgb_to_hsv`` / ``hsv_to_rgb``, tensorflow 1.x), the check TensorFlow's
own unit test applies to them: agreement with Python's ``colorsys`` tuple by
tuple (plus matplotlib as a second implementation) -- see
``tests/test_oracle_*.py`` and DESIGN.md section 1(c).

Since round 5 ONE part is pinned by the reference itself: the forward
arithmetic of Exposure, Gamma (x >= 0.001) and WhiteBalance (regressor
normalisation + process), ``rgb2lum`` and ``lerp`` are checked against vectors
produced by running the reference's own NumPy code (``user_study_ui/filters.py``,
``util.py``) in the build container -- ``tests/golden/make_reference_vectors.py``,
``tests/test_reference_vectors.py``.  Everything that lives in TensorFlow (the
other five filters, every gradient, the tie conventions) stays unpinned.

Since round 6 a COMPOSITION pin covers that TensorFlow side -- TF primitive
semantics ASSUMED: the bodies of ``process`` / ``filter_param_regressor`` of all
nine filter classes, the ``util.py`` helpers, ``pdf_sample`` and the selection /
state / penalty statements of ``agent_generator`` are executed in the build
container under a NumPy facade for ``tf`` (``tests/golden/make_reference_facade.py``)
and the restatements here reproduce them to 1e-12 (``tests/test_reference_facade.py``).
It pins how the primitives are composed, not the primitives: parity stays
"partial" (DESIGN.md section 7).
"""
