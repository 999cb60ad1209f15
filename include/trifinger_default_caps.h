/* trifinger_default_caps.h - collision shapes of the finger links of tf_default_model().  DATA ONLY: the output of
 * `python tools/fit_link_shapes.py {distal,middle,upper}` (rounded to 0.1 mm), fitted to the collision hulls of the reference's link
 * meshes (tests/golden/model.npz) so that no point of a hull surface lies more than 3 mm outside the link's shape;
 * tests/test_model_fixture.py holds that, and states the over-coverage.  Included by both libraries so that they ship the same model.
 * Shape: { a[3], b[3], w1{s=0, s=1}, w2{..}, rho{..}, o1{..}, o2{..} }, sphere: { c[3], radius }; link frame, metres. */
#ifndef TRIFINGER_DEFAULT_CAPS_H_
#define TRIFINGER_DEFAULT_CAPS_H_
/* distal body (lower link + tip link): the D-shaped prism tapering from the joint-3 housing to the fingertip sphere; the housing puck */
#define TF_DEFAULT_SHAPE3 { { 0.0135f, 0.0f, 0.0f }, { 0.0185f, 0.0f, -0.1592f }, \
    { 0.0148f, 0.0102f }, { 0.0197f, 0.0102f }, { 0.0081f, 0.0102f }, { -0.0015f, 0.0f }, { 0.0f, 0.0f } }
#define TF_DEFAULT_SPH3 { { 0.0102f, 0.0f, -0.0007f }, 0.0258f }
/* middle link: the motor housing between the joint-2 housing (top) and the joint-3 housing (bottom) */
#define TF_DEFAULT_SHAPE2 { { 0.026f, -0.003f, -0.012f }, { 0.035f, 0.0f, -0.150f }, \
    { 0.0309f, 0.0161f }, { 0.0245f, 0.0241f }, { 0.0128f, 0.0087f }, { -0.0034f, -0.0024f }, { 0.0008f, 0.0019f } }
#define TF_DEFAULT_SPH2 { { 0.0098f, -0.0030f, -0.0016f }, 0.0262f }, { { 0.0383f, 0.0011f, -0.1606f }, 0.0246f }
/* upper link (only reached by a cube above upper_check_z); width directions x and z */
#define TF_DEFAULT_SHAPE1 { { 0.004f, 0.040f, 0.0f }, { 0.004f, 0.218f, 0.0f }, \
    { 0.0214f, 0.0276f }, { 0.0290f, 0.0276f }, { 0.0186f, 0.0209f }, { -0.0002f, -0.0006f }, { -0.0010f, 0.0008f } }
#endif
