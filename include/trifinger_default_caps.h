/* trifinger_default_caps.h - the capsule table of tf_default_model(): collision shape of the finger links besides the fingertip
 * capsule (TfModel.caps).  DATA ONLY: the output of `python tools/fit_link_capsules.py {distal,middle,upper}` (rounded to 0.1 mm),
 * fitted to the collision hulls of the reference's link meshes (tests/golden/model.npz) so that no point of a hull surface lies more
 * than 3 mm outside the union of the link's capsules; tests/test_model_fixture.py holds that, and states the over-coverage.
 * Included by both libraries so that they ship the same model.  Consecutive entries of one link form a group (bounding-sphere cull).
 * Entry: { link (1 upper, 2 middle, 3 lower), a[3], b[3], radius } in the link frame, metres. */
#ifndef TRIFINGER_DEFAULT_CAPS_H_
#define TRIFINGER_DEFAULT_CAPS_H_
#define TF_DEFAULT_CAPS \
    /* distal body (lower link + tip link): a pair fanning out from the tube to the joint housing, two capsules across the housing */ \
    { 3, { 0.0113f,  0.0123f, -0.0021f }, { 0.0157f,  0.0028f, -0.0976f }, 0.0129f }, \
    { 3, { 0.0113f, -0.0123f, -0.0021f }, { 0.0157f, -0.0028f, -0.0976f }, 0.0129f }, \
    { 3, { 0.0095f, -0.0105f,  0.0097f }, { 0.0095f,  0.0105f,  0.0097f }, 0.0132f }, \
    { 3, { 0.0095f, -0.0105f, -0.0095f }, { 0.0095f,  0.0105f, -0.0095f }, 0.0132f }, \
    /* middle link: 2 x 2 capsules along the body, one across each joint housing */ \
    { 2, { 0.0115f,  0.0089f, -0.0185f }, { 0.0331f,  0.0089f, -0.1624f }, 0.0148f }, \
    { 2, { 0.0115f, -0.0140f, -0.0185f }, { 0.0331f, -0.0140f, -0.1624f }, 0.0148f }, \
    { 2, { 0.0330f,  0.0068f, -0.0185f }, { 0.0393f,  0.0068f, -0.1624f }, 0.0151f }, \
    { 2, { 0.0330f, -0.0118f, -0.0185f }, { 0.0393f, -0.0118f, -0.1624f }, 0.0151f }, \
    { 2, { 0.0178f, -0.0025f, -0.0028f }, { 0.0178f, -0.0025f, -0.0028f }, 0.0294f }, \
    { 2, { 0.0380f,  0.0000f, -0.1679f }, { 0.0337f,  0.0000f, -0.1679f }, 0.0219f }, \
    /* upper link (only reached by a cube above upper_check_z): two pairs along the link */ \
    { 1, { -0.0041f, 0.0390f,  0.0092f }, { -0.0082f, 0.2233f,  0.0092f }, 0.0170f }, \
    { 1, { -0.0041f, 0.0390f, -0.0092f }, { -0.0082f, 0.2233f, -0.0092f }, 0.0170f }, \
    { 1, {  0.0107f, 0.0390f,  0.0076f }, {  0.0146f, 0.2233f,  0.0076f }, 0.0169f }, \
    { 1, {  0.0107f, 0.0390f, -0.0076f }, {  0.0146f, 0.2233f, -0.0076f }, 0.0169f }
#endif
