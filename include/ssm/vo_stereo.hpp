// ssm/vo_stereo.hpp -- only the parameter block of the reference's VisualOdometryStereo (include/vo_stereo.hpp), which
// Tracker's constructor takes by value (track.h:64) and exp_mapping.cpp:21-31 fills.  The libviso2-style stereo
// odometry itself is out of scope.
#pragma once
struct VisualOdometryStereo {
    struct calibration { double f = 1, cu = 0, cv = 0; };
    struct parameters { calibration calib; double base = 1; int ransac_iters = 200; double inlier_threshold = 2.0; bool reweighting = true; };
};
