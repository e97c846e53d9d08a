// test_host -- GPU-side checks of the C++ host classes (run by tests/test_host_cpp.py under -m gpu).
// Prints one "PASS name" / "FAIL name" line per check; exit code = number of failures.
#include "ssm/rgbdframe.h"
#include "ssm/track.h"
#include "ssm/pose_graph.h"
#include "ssm/mapper.h"
#include "ssm/segnet.h"
#include "ssm/quadmatcher.hpp"
#include "ssm/vo_stereo.hpp"
#include "ssm/stereo.h"
#include "ssm/batch_stereo_tracker.h"
#include "ssm/batch_tracker.h"
using namespace std;
using namespace rgbd_tutor;
static int fails = 0;
#define CHECK(name, cond) do { if (cond) cout << "PASS " << name << endl; else { cout << "FAIL " << name << endl; fails++; } } while (0)

int main(int argc, char** argv)
{
    ParameterReader para(argc > 1 ? argv[1] : "./parameters.txt");
    if (argc > 3) {                      // dataset-layout checks only (tests/test_host_cpp.py::test_frame_reader_tum_and_kitti_layouts)
        para.set("data_source", argv[2]); para.set("start_index", "0"); para.set("end_index", "100");
        FrameReader tum(para, FrameReader::TUM);
        RGBDFrame::Ptr a = tum.next(), b = tum.next(), c3 = tum.next(), d = tum.next();
        CHECK("frame_reader_tum", a && b && c3 && !d && a->rgb.cols == 640 && a->rgb.type() == CV_8UC3 && a->depth.type() == CV_16UC1 &&
              a->depth.at<ushort>(100, 100) == 1000 && c3->depth.at<ushort>(5, 5) == 1002 && c3->id == 2);
        para.set("data_source", argv[3]); para.set("camera.baseline", "0.532331858"); para.set("camera.roix", "2000"); para.set("camera.roiy", "2000"); para.set("camera.roiz", "4000");
        FrameReader kitti(para, FrameReader::KITTI);
        RGBDFrame::Ptr k0 = kitti.next(), k1 = kitti.next(), k2 = kitti.next();
        int good = 0; if (k0) for (int y = 10; y < 110; y++) for (int x = 120; x < 390; x++) if (abs(k0->disparity.at<int16_t>(y, x) - 24 * 16) <= 8) good++;
        CHECK("frame_reader_kitti", k0 && k1 && !k2 && k0->img_lc.cols == 400 && k0->img_rp.rows == 120 && k0->depth.type() == CV_16UC1 && good > 26000 && k0->depth.at<ushort>(60, 200) > 0);
        // Tracker in stereo mode (Tracker::estimateVO): the synthetic pair is a static fronto-parallel plane, so the motion is ~identity
        {
            para.set("tracker_mode", "stereo"); para.set("image_width", "400"); para.set("image_height", "120"); para.set("orb_levels", "3"); para.set("orb_features", "300");
            VisualOdometryStereo::parameters vp; vp.calib.f = para.getData<double>("camera.fx"); vp.calib.cu = para.getData<double>("camera.cx"); vp.calib.cv = para.getData<double>("camera.cy");
            vp.base = 0.532331858; vp.inlier_threshold = 2.0;
            Tracker tracker(para, vp);
            FrameReader again(para, FrameReader::KITTI);
            RGBDFrame::Ptr a0 = again.next(), a1 = again.next();
            tracker.updateFrame(a0);
            Eigen::Isometry3d T1 = tracker.updateFrame(a1);
            CHECK("tracker_stereo_mode_runs_estimateVO", a0 && a1 && tracker.getState() != Tracker::NOT_READY && tracker.lastMatches >= 0 && std::isfinite(T1(0, 3)));
        }
        // the stereo tracker in bulk against the per-frame classes on a coherent sequence (argv[4]: a textured plane moving by 2 px per image, one jump of 150 px
        // in the middle): with tracker_max_lost_frame = 0 the jump makes the tracker LOST, the frame behind it goes through lostRecover although its quad
        // matcher finds matches -- the case in which the bulk call's rand() stream has to be put right (include/ssm/batch_stereo_tracker.h)
        if (argc > 4) {
            para.set("data_source", argv[4]); para.set("tracker_max_lost_frame", "0"); para.set("end_index", "100");
            VisualOdometryStereo::parameters vp; vp.calib.f = para.getData<double>("camera.fx"); vp.calib.cu = para.getData<double>("camera.cx"); vp.calib.cv = para.getData<double>("camera.cy");
            vp.base = 0.532331858; vp.inlier_threshold = 2.0;
            vector<Eigen::Isometry3d> Ta; vector<cv::Mat> Da; vector<int> Sa;
            {
                para.set("kitti_reader_depth", "1");
                Tracker tracker(para, vp); FrameReader rd(para, FrameReader::KITTI);
                while (RGBDFrame::Ptr f = rd.next()) { tracker.updateFrame(f); Ta.push_back(f->getTransform()); Da.push_back(f->depth); Sa.push_back((int)tracker.getState()); }
            }
            // chunks of 3 (the LOST frame opens a chunk, the recovering frame sits inside it), 4 (the recovering frame opens a chunk) and 8 (one chunk)
            bool all_chunks = true;
            for (int chunk : {3, 4, 8}) {
            vector<Eigen::Isometry3d> Tb; vector<cv::Mat> Db; vector<int> Sb; int tracked = 0;
            {
                para.set("kitti_reader_depth", "0");
                FrameReader rd(para, FrameReader::KITTI);
                BatchStereoTracker bs(para, vp, 400, 120, chunk);
                auto take = [&](const vector<RGBDFrame::Ptr>& done) { for (size_t i = 0; i < done.size(); i++) { Tb.push_back(done[i]->getTransform()); Db.push_back(done[i]->depth); Sb.push_back(bs.infos[i].state); tracked += bs.infos[i].tracked; } };
                while (RGBDFrame::Ptr f = rd.next()) take(bs.push(f));
                take(bs.flush());
                para.set("kitti_reader_depth", "1");
            }
            bool same = Ta.size() == Tb.size() && Ta.size() == 8;
            int lost = 0; double moved = 0;
            for (size_t i = 0; same && i < Ta.size(); i++) {
                same = memcmp(Ta[i].matrix().data(), Tb[i].matrix().data(), 128) == 0 && Sa[i] == Sb[i] && Da[i].rows == Db[i].rows && memcmp(Da[i].data, Db[i].data, (size_t)Da[i].rows * Da[i].cols * 2) == 0;
                lost += Sa[i] == Tracker::LOST; moved = max(moved, fabs(Ta[i](0, 3)));
            }
            if (!(same && lost >= 1 && tracked >= 5 && moved > 1e-3)) {
                cout << "  chunk " << chunk << " frames " << Ta.size() << " / " << Tb.size() << " lost " << lost << " tracked " << tracked << " moved " << moved << endl;
                for (size_t i = 0; i < Ta.size() && i < Tb.size(); i++)
                    cout << "  frame " << i << ": state " << Sa[i] << " / " << Sb[i] << " tx " << Ta[i](0, 3) << " / " << Tb[i](0, 3) << " pose equal " << (memcmp(Ta[i].matrix().data(), Tb[i].matrix().data(), 128) == 0)
                         << " depth equal " << (Da[i].rows == Db[i].rows && Da[i].rows > 0 && memcmp(Da[i].data, Db[i].data, (size_t)Da[i].rows * Da[i].cols * 2) == 0)
                         << " (" << Da[i].rows << "x" << Da[i].cols << " type " << Da[i].type() << " / " << Db[i].rows << "x" << Db[i].cols << " type " << Db[i].type() << "; [60,200] " << (Da[i].rows ? Da[i].at<ushort>(60, 200) : 0) << " / " << (Db[i].rows ? Db[i].at<ushort>(60, 200) : 0) << ")" << endl;
            }
            all_chunks = all_chunks && same && lost >= 1 && tracked >= 5 && moved > 1e-3;
            }
            CHECK("bulk_stereo_tracker_equals_per_frame_tracker", all_chunks);
        }
        cout << (fails ? "FAILED" : "ALL PASSED") << endl;
        return fails;
    }
    CHECK("parameter_reader", para.getData<int>("orb_features") == 1000 && para.getData<double>("knn_match_ratio") == 0.8 && !para.has("#comment"));
    bool threw = false; try { para.getData<int>("no_such_key"); } catch (const out_of_range&) { threw = true; }
    CHECK("parameter_missing_key_throws", threw);

    FrameReader reader(para, FrameReader::SYNTHETIC);
    RGBDFrame::Ptr f0 = reader.next(), f1 = reader.next();
    CHECK("frame_reader", f0 && f1 && f0->id == 0 && f1->id == 1 && f0->rgb.cols == 640 && f0->depth.rows == 480 && f1->T_f_w(0, 3) == 0.01);

    OrbFeature orb(para);
    orb.detectFeatures(f0); orb.detectFeatures(f1);
    CHECK("detectFeatures", f0->features.size() > 900 && f0->features.size() <= 1024 && f0->features[0].descriptor.cols == 32);
    bool pos_ok = true;
    for (auto& ft : f0->features) { cv::Point3f p = f0->project2dTo3d((int)ft.keypoint.pt.x, (int)ft.keypoint.pt.y); if (!(p == ft.position)) pos_ok = false; }
    CHECK("feature_position_is_project2dTo3d", pos_ok);
    vector<cv::DMatch> m = orb.match(f0, f1);
    bool asc = true; for (size_t i = 1; i < m.size(); i++) if (m[i].queryIdx <= m[i-1].queryIdx) asc = false;
    CHECK("match", m.size() > 200 && asc && m[0].imgIdx == 0);
    vector<cv::DMatch> self = orb.match(f0, f0);
    bool selfok = self.size() > 0; for (auto& d : self) if (d.queryIdx != d.trainIdx || d.distance != 0) selfok = false;
    CHECK("match_self_is_identity", selfok);

    // PnP on exact synthetic correspondences: recover a known world->camera transform
    {
        PnPSolver pnp(para, orb);
        Eigen::Isometry3d Tgt; const double a = 0.05;
        Tgt(0, 0) = cos(a); Tgt(0, 2) = sin(a); Tgt(2, 0) = -sin(a); Tgt(2, 2) = cos(a); Tgt(0, 3) = 0.03; Tgt(1, 3) = -0.02; Tgt(2, 3) = 0.05;
        vector<cv::Point3f> obj; vector<cv::Point2f> img; CAMERA_INTRINSIC_PARAMETERS k = para.getCamera();
        unsigned s = 12345;
        for (int i = 0; i < 200; i++) {
            auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
            cv::Point3f X((float)(rnd() * 2 - 1), (float)(rnd() * 1.5 - 0.75), (float)(1.0 + rnd() * 2));
            Eigen::Vector4d p = Tgt * Eigen::Vector4d(X.x, X.y, X.z, 1);
            cv::Point2f u((float)(k.fx * p(0) / p(2) + k.cx), (float)(k.fy * p(1) / p(2) + k.cy));
            if (i % 10 == 0) { u.x += 40; u.y -= 25; }                       // 10 % gross outliers
            obj.push_back(X); img.push_back(u);
        }
        vector<int> inl; Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
        bool ok = pnp.solvePnP(img, obj, k, inl, T);
        double err = 0; for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) err = max(err, fabs(T(r, c) - Tgt(r, c)));
        CHECK("pnp_recovers_pose_and_rejects_outliers", ok && err < 1e-3 && inl.size() == 180);
    }

    // VisualOdometryStereo on synthetic quad matches (the consumer of QuadFeatureMatch::quadmatches, src/track.cpp:57-66)
    {
        VisualOdometryStereo::parameters vp; vp.calib.f = 718.856; vp.calib.cu = 607.1928; vp.calib.cv = 185.2157; vp.base = 0.5323; vp.inlier_threshold = 2.0;
        VisualOdometryStereo viso(vp);
        QuadFeatureMatch qm;
        const double tr[6] = {0.01, -0.02, 0.005, 0.05, -0.02, -0.8};
        const double sx = sin(tr[0]), cx = cos(tr[0]), sy = sin(tr[1]), cy = cos(tr[1]), sz = sin(tr[2]), cz = cos(tr[2]);
        const double R[9] = {cy*cz, -cy*sz, sy, sx*sy*cz+cx*sz, -sx*sy*sz+cx*cz, -sx*cy, -cx*sy*cz+sx*sz, cx*sy*sz+sx*cz, cx*cy};
        unsigned s = 777;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
        for (int i = 0; i < 300; i++) {
            const double X = rnd() * 20 - 10, Y = rnd() * 4 - 2, Z = 5 + rnd() * 35;
            const double Xc = R[0]*X + R[1]*Y + R[2]*Z + tr[3], Yc = R[3]*X + R[4]*Y + R[5]*Z + tr[4], Zc = R[6]*X + R[7]*Y + R[8]*Z + tr[5];
            pmatch q; memset(static_cast<void*>(&q), 0, sizeof(q));
            q.u1p = (float)(vp.calib.f * X / Z + vp.calib.cu); q.v1p = (float)(vp.calib.f * Y / Z + vp.calib.cv); q.u2p = (float)(vp.calib.f * (X - vp.base) / Z + vp.calib.cu); q.v2p = q.v1p;
            q.u1c = (float)(vp.calib.f * Xc / Zc + vp.calib.cu); q.v1c = (float)(vp.calib.f * Yc / Zc + vp.calib.cv); q.u2c = (float)(vp.calib.f * (Xc - vp.base) / Zc + vp.calib.cu); q.v2c = q.v1c;
            if (i % 6 == 0) { q.u1c += 35; q.v2c -= 18; }                    // gross outliers
            qm.quadmatches.push_back(q);
        }
        const bool ok = viso.Process(qm);
        cv::Mat M = viso.getMotion();
        double err = 0;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) err = max(err, fabs(M.at<double>(r, c) - R[r * 3 + c])); err = max(err, fabs(M.at<double>(r, 3) - tr[3 + r])); }
        CHECK("vo_stereo_recovers_motion", ok && err < 1e-3 && viso.getNumberOfMatches() == 300 && viso.getNumberOfInliers() == 250 &&
              viso.quadmatches_inlier.size() == 250 && viso.quadmatches_outlier.size() == 50);
        QuadFeatureMatch few; few.quadmatches.assign(qm.quadmatches.begin(), qm.quadmatches.begin() + 5);
        CHECK("vo_stereo_needs_six_matches", !viso.Process(few));
    }

    // calDisparity_SGBM + the depth conversion on a synthetic rectified pair (a textured plane at disparity 24)
    {
        const int W = 320, H = 96, D0 = 24;
        cv::Mat L(H, W, CV_8UC1), R(H, W, CV_8UC1);
        unsigned s = 4242;
        std::vector<uint8_t> tex((size_t)H * (W + 64));
        for (auto& v : tex) { s = s * 1664525u + 1013904223u; v = (uint8_t)(s >> 24); }
        auto T = [&](int y, int x) { int a = 0; for (int k = -1; k <= 1; k++) a += tex[(size_t)y * (W + 64) + x + 1 + k]; return (uint8_t)(a / 3); };
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { R.at<uchar>(y, x) = T(y, x + 30); L.at<uchar>(y, x) = T(y, x + 30 - D0); }
        cv::Mat disp, depth, disp2;
        calDisparity_SGBM(L, R, disp);
        int good = 0, valid = 0;
        for (int y = 8; y < H - 8; y++) for (int x = 100; x < W - 8; x++) { const int d = disp.at<int16_t>(y, x); if (d != -16) { valid++; if (abs(d - D0 * 16) <= 8) good++; } }
        CHECK("calDisparity_SGBM_recovers_a_plane", disp.rows == H && disp.cols == W && valid > 10000 && good > valid * 0.98 && disp.at<int16_t>(10, 5) == -16);
        stereoDepth(L, R, 0.532331858, 607.1928 - 450, 185.2157 - 140, 718.856, 20, 5, 40, 1000.0, depth, disp2);
        const int zexp = (int)(718.856 * (0.532331858 / (double)(D0 * 16)) * 16.0 * 1000.0);
        bool same = true; for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) if (disp.at<int16_t>(y, x) != disp2.at<int16_t>(y, x)) same = false;
        int zok = 0; for (int y = 20; y < H - 20; y++) for (int x = 120; x < 200; x++) if (depth.at<ushort>(y, x) == zexp) zok++;
        CHECK("stereoDepth_matches_the_reference_formula", same && zok > 2000);
    }

    // Tracker state machine on a static scene (the same frame fed three times): state OK, pose ~ identity (not exact: 3-D positions come from truncated pixel coordinates, include/orb.h:50, while the 2-D side is the scaled sub-pixel keypoint)
    {
        VisualOdometryStereo::parameters vo;
        Tracker tracker(para, vo);
        RGBDFrame::Ptr a = reader.get(5), b = reader.get(5), c = reader.get(5);
        a->T_f_w = Eigen::Isometry3d::Identity(); b->T_f_w = a->T_f_w; c->T_f_w = a->T_f_w;
        tracker.updateFrame(a);
        CHECK("tracker_first_frame", tracker.getState() == Tracker::OK && tracker.referenceFrames().size() == 1);
        Eigen::Isometry3d T1 = tracker.updateFrame(b); tracker.updateFrame(c);
        double dev = 0; for (int r = 0; r < 3; r++) for (int cc = 0; cc < 4; cc++) dev = max(dev, fabs(T1(r, cc) - (r == cc ? 1.0 : 0.0)));
        cout << "  static scene: state " << tracker.getState() << " dev " << dev << " matches " << tracker.lastMatches << " inliers " << tracker.lastInliers << " refs " << tracker.referenceFrames().size() << endl;
        CHECK("tracker_static_scene", tracker.getState() == Tracker::OK && dev < 5e-3 && tracker.lastInliers > 100 && tracker.referenceFrames().size() == 3);
    }

    // The bulk pose chain (ssm_seq_process + ssm_tracker_run behind BatchTracker) against the per-frame Tracker: 50 frames of the synthetic stream with
    // three flat frames inside.  One flat frame fails to track (the deque then holds a frame older than the match-table window: matched on demand);
    // two in a row exceed tracker_max_lost_frame = 1 -> LOST -> lostRecover on the next frame.  Every T_f_w must be the same bits.
    {
        ParameterReader pt = para; pt.set("tracker_max_lost_frame", "1"); pt.set("ssm_max_batch", "8");
        const int NF = 50;
        auto make = [&](int i) { RGBDFrame::Ptr f = reader.get(i); if (i == 12 || i == 30 || i == 31) { memset(f->rgb.data, 100, (size_t)f->rgb.rows * f->rgb.step); } return f; };
        VisualOdometryStereo::parameters vo;
        Tracker per(pt, vo);
        vector<Eigen::Isometry3d> Tper; vector<int> st_per;
        for (int i = 0; i < NF; i++) { RGBDFrame::Ptr f = make(i); per.updateFrame(f); Tper.push_back(f->getTransform()); st_per.push_back((int)per.getState()); }
        RGBDFrame::Ptr f00 = make(0);
        BatchTracker bulk(pt, f00->rgb.cols, f00->rgb.rows, f00->T_f_w, 17);      // chunks of 17: the chain crosses chunk borders (and sub-batches of 8 inside)
        vector<RGBDFrame::Ptr> done; vector<ssm_track_info> infos;
        for (int i = 0; i < NF; i++) { for (auto& f : bulk.push(make(i))) done.push_back(f); if (done.size() > infos.size()) infos.insert(infos.end(), bulk.infos.begin(), bulk.infos.end()); }
        for (auto& f : bulk.flush()) done.push_back(f);
        if (done.size() > infos.size()) infos.insert(infos.end(), bulk.infos.begin(), bulk.infos.end());
        bool same = done.size() == (size_t)NF && infos.size() == (size_t)NF; int first_bad = -1, lost = 0, untracked = 0;
        for (int i = 0; same && i < NF; i++) {
            if (memcmp(done[i]->getTransform().data(), Tper[i].data(), 128) != 0 || infos[i].state != st_per[i]) { same = false; first_bad = i; }
            lost += infos[i].state == 2; untracked += !infos[i].tracked;
        }
        double drift = 0; if (same) drift = fabs(Tper[NF - 1](0, 3));
        cout << "  bulk tracker: first mismatch " << first_bad << " lost-frames " << lost << " untracked " << untracked << " |tx| of the last pose " << drift << endl;
        // (the synthetic stream is not a rigid scene -- its depth pattern stays put while the texture pans -- so PnP also fails on its own now and then:
        // more LOST / lostRecover events than the three flat frames force, all of them compared)
        CHECK("bulk_tracker_equals_per_frame_tracker", same && lost >= 1 && untracked >= 3 && !infos[12].tracked && !infos[30].tracked && !infos[31].tracked);
        // the same chain with the regular stretches solved on the GPU (one block of 1024 threads per frame run: kernels_pnp.hip): still the same bits
        pt.set("tracker_pnp_on_device", "1");
        BatchTracker bulkd(pt, f00->rgb.cols, f00->rgb.rows, f00->T_f_w, 23);
        vector<RGBDFrame::Ptr> dd;
        for (int i = 0; i < NF; i++) for (auto& f : bulkd.push(make(i))) dd.push_back(f);
        for (auto& f : bulkd.flush()) dd.push_back(f);
        bool same_d = dd.size() == (size_t)NF; int bad_d = -1;
        for (int i = 0; same_d && i < NF; i++) if (memcmp(dd[i]->getTransform().data(), Tper[i].data(), 128) != 0) { same_d = false; bad_d = i; }
        cout << "  device chain: first mismatch " << bad_d << endl;
        CHECK("device_pose_chain_equals_per_frame_tracker", same_d);
    }

    // PoseGraph key-frame gate + Mapper viewer thread
    {
        VisualOdometryStereo::parameters vo;
        Tracker::Ptr tracker(new Tracker(para, vo));
        PoseGraph pg(para, tracker);
        Mapper mapper(para, pg);
        reader.reset();
        int inserted = 0;
        for (int i = 0; i < 8; i++) { RGBDFrame::Ptr f = reader.next(); inserted += pg.tryInsertKeyFrame(f); }
        CHECK("keyframe_gate", inserted == 8 && pg.keyframes.size() == 8);     // test parameters: keyframe_min_translation below the 0.01 m step
        for (int k = 0; k < 300 && mapper.updates() < 1; k++) this_thread::sleep_for(chrono::milliseconds(10));
        this_thread::sleep_for(chrono::milliseconds(200));
        pg.shutdown(); mapper.shutdown();
        Mapper::PointCloud::Ptr gm = mapper.getGlobalMap();
        bool sorted_ok = gm && gm->points.size() > 500;
        CHECK("mapper_viewer_builds_map", sorted_ok && mapper.updates() >= 1);
        // the map of one update == VoxelGrid over the clouds it added (first update: every 2nd key-frame)
        Mapper::PointCloud::Ptr c0 = mapper.generatePointCloud(pg.keyframes[0]);
        CHECK("generatePointCloud", c0->points.size() > 100000 && c0->points[0].data3 == 1.0f);
        {   // frame->pointcloud is cached like the reference (src/mapper.cpp:17-20): a second call with another pose re-uses it, and the host-side
            // pcl::transformPointCloud gives the bits the device gives when ssm_backproject is handed the pose
            const int before = mapper.cloudsComputed;
            RGBDFrame::Ptr kf = pg.keyframes[0];
            Eigen::Isometry3d T2 = Eigen::Isometry3d::Identity();
            T2(0, 0) = 0.8; T2(0, 1) = -0.6; T2(1, 0) = 0.6; T2(1, 1) = 0.8; T2(0, 3) = 0.123456789; T2(1, 3) = -3.25; T2(2, 3) = 1.0 / 3.0;
            kf->setTransform(T2);
            Mapper::PointCloud::Ptr c1 = mapper.generatePointCloud(kf);
            ssm_config cfg = para.deviceConfig(kf->depth.cols, kf->depth.rows); ssm::Device dev(cfg);
            ssm_camera cam; cam.cx = kf->camera.cx; cam.cy = kf->camera.cy; cam.fx = kf->camera.fx; cam.fy = kf->camera.fy; cam.scale = kf->camera.scale;
            vector<ssm_point> ref((size_t)kf->depth.cols * kf->depth.rows); int n = 0;
            dev.check(ssm_backproject(dev.ctx(), kf->depth.ptr<uint16_t>(), kf->rgb.data, kf->semantic.data, kf->depth.cols, kf->depth.rows, &cam, T2.data(),
                                      para.getData<double>("mapper_max_distance", 40.0), ref.data(), (int)ref.size(), &n), "ssm_backproject");
            bool same = (size_t)n == c1->points.size() && mapper.cloudsComputed == before && kf->pointcloud != nullptr;
            for (int i = 0; same && i < n; i++) if (memcmp(&ref[i], &c1->points[i], 16) != 0 || ref[i].r != c1->points[i].r || ref[i].label != c1->points[i].label) same = false;
            CHECK("generatePointCloud_cached_and_transformed_like_device", same);
        }
        ifstream pcd(para.getData<string>("map_output"), ios::binary); string line; getline(pcd, line);
        CHECK("pcd_written", (bool)pcd && line.find(".PCD") != string::npos);
    }
    // QuadFeatureMatch on a synthetic rectified stereo pair: right = left shifted by the disparity, previous = current shifted by the flow
    {
        const int W = 640, H = 480;
        auto gray_of = [&](const cv::Mat& bgr) { cv::Mat g(H, W, CV_8UC1); for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const uchar* p = bgr.ptr<uchar>(y) + 3 * x; g.ptr<uchar>(y)[x] = (uchar)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14); } return g; };
        auto shift = [&](const cv::Mat& a, int dx, int dy) { cv::Mat o(H, W, CV_8UC1); for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) o.ptr<uchar>(y)[x] = a.ptr<uchar>(((y - dy) % H + H) % H)[((x - dx) % W + W) % W]; return o; };
        cv::Mat lc = gray_of(f0->rgb), rc = shift(lc, -9, 0), lp = shift(lc, 2, 1), rp = shift(lp, -9, 0), s1, s2;
        QuadFeatureMatch* qm = new QuadFeatureMatch(lc, rc, lp, rp, s1, s2, true);
        qm->init(DET_GFTT, DES_SIFT);                      // exactly the call of Tracker::estimateVO (track.cpp:52)
        qm->detectFeature();
        qm->circularMatching();
        bool ok = qm->quadmatches.size() > 300;
        double md = 0; for (auto& m : qm->quadmatches) md += (m.u1c - m.u2c); md /= max<size_t>(qm->quadmatches.size(), 1);
        CHECK("quadmatcher_tracking_branch", ok && fabs(md - 9.0) < 0.2);
        delete qm;
        QuadFeatureMatch qb(lc, rc, lp, rp, s1, s2, false);
        qb.init(DET_ORB, DES_ORB); qb.detectFeature(); qb.circularMatching();
        bool ok2 = qb.quadmatches.size() > 50; for (auto& m : qb.quadmatches) if (!(fabs(m.u1c - m.u2c) > 3)) ok2 = false;
        CHECK("quadmatcher_matching_branch", ok2);
    }
    // Classifier (SegNet) from a weight file, when the test harness provides one
    if (argc > 2) {
        Classifier classifier(argv[2], "/nonexistent/semantic12.txt");
        std::vector<Prediction> pr = classifier.Classify(f0->rgb);
        bool ok = pr.size() == 360u * 480u; int mx = 0; for (auto& q : pr) { if (q.second < 0 || q.second > 11) ok = false; mx = max(mx, q.second); }
        cv::Mat sem = classifier.ColorLabels(f0->rgb);
        CHECK("classifier_classify", ok && mx > 0 && sem.rows == 480 && sem.cols == 640 && sem.channels() == 3);
        bool threw2 = false; try { Classifier bad("/nonexistent.ssmw", "x"); } catch (const runtime_error&) { threw2 = true; }
        CHECK("classifier_missing_model_throws", threw2);
        // the reference's constructor arguments (src/segnet.cpp:17-19): prototxt + caffemodel + label file; the harness wrote the SAME weights
        // as a .caffemodel (conv bias + caffe-segnet BN blobs), so the labels must be identical
        if (const char* cmf = getenv("SSM_TEST_CAFFEMODEL")) {
            Classifier from_caffe("", cmf, "/nonexistent/semantic12.txt");
            std::vector<Prediction> pc = from_caffe.Classify(f0->rgb);
            bool same = pc.size() == pr.size(); for (size_t i = 0; same && i < pc.size(); i++) if (pc[i].second != pr[i].second || pc[i].first != pr[i].first) same = false;
            CHECK("classifier_from_caffemodel_equals_ssmw", same);
            bool threw3 = false; try { Classifier bad("/nonexistent.prototxt", cmf, "x"); } catch (const runtime_error&) { threw3 = true; }
            CHECK("classifier_missing_prototxt_throws", threw3);
        }
    }
    cout << (fails ? "FAILED " : "ALL PASSED ") << fails << endl;
    return fails;
}
