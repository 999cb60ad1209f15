/* trifinger_default_caps.h - the capsule table of tf_default_model(): collision shape of the finger links besides the fingertip
 * capsule (TfModel.caps).  DATA ONLY, written by `python tools/fit_link_capsules.py --write` from the collision hulls of the
 * reference's link meshes (tests/golden/model.npz); included by both libraries so that they ship the same model.
 * Entry: { link (1 upper, 2 middle, 3 lower), a[3], b[3], radius } in the link frame, metres. */
#ifndef TRIFINGER_DEFAULT_CAPS_H_
#define TRIFINGER_DEFAULT_CAPS_H_
#define TF_DEFAULT_CAPS \
    { 2, { 0.028f, 0.0f, 0.0f }, { 0.028f, 0.0f, -0.16f }, 0.022f }, \
    { 1, { 0.005f, 0.045f, 0.0f }, { 0.005f, 0.21f, 0.0f }, 0.024f }
#endif
