"""exposure_amd -- Exposure's differentiable filter stack on MI355X (gfx950): HIP kernels behind a C-ABI
(include/exposure_hip.h) under the reference's Filter / agent / critic Python surface.  DESIGN.md has the map.

(Rounds 4-5 shipped measured MIOpen solver rankings and pointed MIOpen at a private copy of them here; since round 6
every convolution of the training iteration runs on the in-house kernels of csrc/conv_ops.hip, so the package no longer
touches MIOpen's environment.)"""
