// experiments.h -- the fence between the product library and the laboratory.
//
// The product build (no -DOHS_EXPERIMENTS) reads no environment variable, carries no alternative kernel generation and
// honours none of the compile-time variants below: defining one without OHS_EXPERIMENTS is a build error.  The
// experiments build (`python -m open_headstage_amd.build --experiments` -> libohs_hip_exp.so, or OHS_BUILD_TAG /
// OHS_EXTRA_DEFS variants of it) adds the environment knobs, the debug entry points (ohs_debug_set_tuning, stamps) and
// round 2's EQ kernel for A/B runs; LABNOTES.md lists them.  Knock-out variants that computed wrong results on
// purpose (round 2-3 timing experiments) are gone from the tree; git history and LABNOTES.md keep their numbers.
// Code that exists only for the experiments build -- round 2's EQ ring kernel and its generated asm, the per-wave stamp
// collectors -- lives in csrc/experiments/ and is included only under OHS_EXPERIMENTS; the natural-layout and
// plain-access variants of k_conv_p1 (measured, slower: LABNOTES.md) were removed in round 5.
#pragma once

#ifndef OHS_EXPERIMENTS
#if defined(OHS_P1_STAMPS) || defined(OHS_EQ_STAMPS) || defined(OHS_P1_WAVES) || defined(OHS_TP_WAVES) || \
    defined(OHS_MAC_TB) || defined(OHS_MAC_PI)
#error "compile-time experiment variants need -DOHS_EXPERIMENTS (python -m open_headstage_amd.build --experiments)"
#endif
#endif
