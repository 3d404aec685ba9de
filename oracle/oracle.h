/* oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * The oracle is a dependency-free, single-threaded CPU restatement of the RGB-L front-end hot path of
 * TUMFTM/ORB_SLAM3_RGBL (ORBextractor, DepthModule, ORBmatcher Hamming paths) plus the OpenCV 4.x
 * primitives those classes delegate to.  It exists to CHECK the HIP path and to be timed as the
 * `cpu_baseline` of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product library (librgbl_frontend.so) never links or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path and cannot be
 * built here (needs OpenCV/Eigen/Pangolin/Boost; none installed, no network).  The oracle is pinned only
 * by (a) hand-derived known-answer tests (tests/test_oracle_kat.py), (b) the live glibc for sinf/cosf
 * and (c) oracle/_ref: the reference's own ORBextractor.cc, DepthModule.cc and ORBmatcher.cc compiled unmodified
 * against minimal stand-ins for OpenCV / Eigen / Sophus / the SLAM data classes (pins everything that is NOT
 * library-internal: cell loop, quad-tree, orientation, steering, packing; settings parsing, projection loop,
 * up-sampling chains, keypoint depth; the matchers' candidate order, masks, thresholds and tie rules).
 */
#ifndef RGBL_ORACLE_H
#define RGBL_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} orc_keypoint;

typedef struct orc_extractor orc_extractor;

/* ORBextractor::ORBextractor (/root/reference/src/ORBextractor.cc:409-469). */
orc_extractor* orc_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                                    int min_th_fast);
void orc_extractor_destroy(orc_extractor*);
/* per-level tables; out arrays hold nlevels entries */
void orc_extractor_tables(const orc_extractor*, float* scale, float* inv_scale, float* sigma2,
                          float* inv_sigma2, int* features_per_level, int* umax16);

/* ORBextractor::operator() (/root/reference/src/ORBextractor.cc:1086-1168).
 * Returns monoIndex, or -1 when the image is empty. *n_out receives the keypoint count (may exceed
 * cap: then only the first cap records are written and -2 is returned). */
int orc_extract(orc_extractor*, const uint8_t* img, int w, int h, int stride, int lap0, int lap1,
                orc_keypoint* kps, uint8_t* desc, int cap, int* n_out);

/* Intermediates of the LAST orc_extract call (for stage-by-stage parity checks). */
int orc_level_size(const orc_extractor*, int level, int* w, int* h);
void orc_level_image(const orc_extractor*, int level, uint8_t* dst, int dst_stride);   /* un-bordered */
void orc_level_blurred(const orc_extractor*, int level, uint8_t* dst, int dst_stride); /* valid if level has keypoints */
/* level image with the 19-px BORDER_REFLECT_101 frame, as mvImagePyramid's parent buffer holds it */
void orc_level_bordered(const orc_extractor*, int level, uint8_t* dst, int dst_stride);
/* FAST candidates handed to DistributeOctTree (coordinates relative to minBorder, CPU order) */
int orc_level_candidates(const orc_extractor*, int level, orc_keypoint* out, int cap);
/* keypoints per level after distribution+orientation (level coordinates, before scaling) */
int orc_level_keypoints(const orc_extractor*, int level, orc_keypoint* out, int cap);

/* Frame::ComputeStereoMatches (/root/reference/src/Frame.cc:901-1071): left/right are extractors whose LAST
 * orc_extract call saw the left/right image (their pyramids are read like mvImagePyramid).  Fills mvuRight /
 * mvDepth (n_left floats each, -1 = no match). */
void orc_stereo_matches(const orc_extractor* left, const orc_extractor* right, const orc_keypoint* kp_left,
                        const uint8_t* desc_left, int n_left, const orc_keypoint* kp_right,
                        const uint8_t* desc_right, int n_right, float mb, float mbf, float* out_uright,
                        float* out_depth);

/* ---- stand-alone OpenCV-semantics primitives ---- */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                          int dstride);
void orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
/* cv::FAST(img, kps, threshold, nonmax) TYPE_9_16; returns count */
int orc_fast(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax, orc_keypoint* out,
             int cap);
/* cornerScore<16> with the given seed threshold at pixel (x,y) (needs a 3-px margin) */
int orc_fast_corner_score(const uint8_t* img, int stride, int x, int y, int threshold);
float orc_fast_atan2(float y, float x);
int orc_cv_round_f(float v);
float orc_ic_angle(const uint8_t* img, int stride, int x, int y);
void orc_brief(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t* desc32);
/* DistributeOctTree on an explicit candidate list; returns count written to out */
int orc_distribute_octree(const orc_keypoint* cand, int n, int min_x, int max_x, int min_y, int max_y,
                          int n_features, orc_keypoint* out, int cap);

/* ---- DBoW2 vocabulary descent: TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
 * (/root/reference/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1208-1255) with FORB::distance (FORB.cpp:79-98), as
 * Frame::ComputeBoW uses it (src/Frame.cc:828-835, levelsup = 4).  The tree as flat arrays: node 0 is the root, the children
 * of node i are child[child_off[i] .. child_off[i+1]) in the order of the vocabulary file, a leaf has no children. */
typedef struct {
  int n_nodes, L;               /* m_nodes.size(), depth levels m_L */
  const int32_t* child_off;     /* n_nodes + 1 */
  const int32_t* child;         /* n_nodes - 1 */
  const uint8_t* desc;          /* 32 bytes per node */
  const double* weight;         /* Node::weight (WordValue) */
  const int32_t* word_id;       /* Node::word_id of the leaves */
} orc_vocabulary;
/* per feature: word id, weight of the word, node id at level L - levelsup (0 = root when that level is <= 0) */
void orc_bow_descend(const orc_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word, double* weight,
                     int32_t* node);

/* ---- ingest either side of the path ---- */
/* cv::cvtColor 8-bit {RGB,BGR,RGBA,BGRA} -> gray, OpenCV 4.x 15-bit weights (Tracking.cc:1567-1580) */
void orc_cvt_gray(const uint8_t* src, int channels, int blue_first, int w, int h, int sstride, uint8_t* dst, int dstride);
/* LoadPointcloudBinaryMat (Examples/RGB-L/rgbl_kitti.cc:151-185): n x (x,y,z,r) -> 4 x n rows x,y,z,1 */
void orc_kitti_bin_to_cloud(const float* xyzi, int n, float* cloud4xn);

/* ---- DepthModule (/root/reference/src/DepthModule.cc:50-274) ---- */
enum { ORC_UPS_NONE = 0, ORC_UPS_NEAREST = 1, ORC_UPS_AVERAGE = 2, ORC_UPS_INVDIL = 3 };
typedef struct {
  float proj[12];     /* LidarProjectionMatrix 3x4 row-major */
  float min_dist, max_dist, mbf;
  int method;         /* ORC_UPS_* */
  int kw, kh;         /* inverse dilation structuring element size */
  uint8_t kernel[81]; /* kh x kw mask, row-major */
  int avg_ksize;      /* AverageFiltering.KernelSize */
  float nn_radius;    /* NearestNeighborPixel.SearchDistance */
} orc_depth_params;

/* cloud: 4 x n row-major (rows x,y,z,1) with leading dimension ld floats.
 * kp_xy: k pairs (x,y) of mvKeys; kpun_x: k values of mvKeysUn.pt.x.
 * out_raw/out_processed may be NULL; each is h*w floats. */
int orc_depth(const orc_depth_params*, const float* cloud, int n, int ld, int w, int h,
              const float* kp_xy, const float* kpun_x, int k, float* out_depth, float* out_uright,
              float* out_raw, float* out_processed);
/* K[3x4] * Tr[4x4] in OpenCV gemm arithmetic (DepthModule.cc:434) */
void orc_projection_matrix(const float K[12], const float Tr[16], float out[12]);
/* cv::distanceTransform(src u8, dst f32, labels, DIST_L2, DIST_MASK_5): distance to the nearest ZERO pixel of src */
void orc_distance_transform_l2_5x5(const uint8_t* src, int w, int h, int stride, float* dst);
/* cv::getStructuringElement for RECT(0)/CROSS(1)/ELLIPSE(2) and the reference's Diamond(3) tables */
int orc_structuring_element(int shape, int kw, int kh, uint8_t* out);

/* ---- ORBmatcher ---- */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b); /* ORBmatcher.cc:2058-2074 */
/* brute force best / second-best (rule of ORBmatcher.cc:283-304: strict '<', first wins) */
/* Frame::ComputeStereoFishEyeMatches (/root/reference/src/Frame.cc:1256-1296) up to the triangulation: knnMatch(k = 2) of
 * the lapping-area subsets [mono, n) + Lowe's ratio 0.7; outputs have n_left entries (-1 / 256 = none). */
void orc_stereo_fisheye_matches(const uint8_t* desc_left, int n_left, int mono_left, const uint8_t* desc_right, int n_right,
                                int mono_right, int* left_to_right, int* best_dist, int* second_dist);
void orc_hamming_bf(const uint8_t* a, int na, const uint8_t* b, int nb, int* best_idx, int* best_dist,
                    int* second_dist);

typedef struct {
  /* key-frame 1 and 2, flat */
  int n1, n2;
  const uint8_t *desc1, *desc2;        /* n x 32 */
  const float *kp1_xy, *kp2_xy;        /* n x 2 (undistorted keypoints) */
  const int *kp1_octave, *kp2_octave;
  const float *kp1_angle, *kp2_angle;
  const float *uright1, *uright2;      /* mvuRight */
  const uint8_t *has_mp1, *has_mp2;    /* MapPoint present */
  /* FeatureVector as CSR: node ids ascending, offsets (nnodes+1), feature indices */
  int nnodes1, nnodes2;
  const int *node_id1, *node_off1, *node_feat1;
  const int *node_id2, *node_off2, *node_feat2;
  float F12[9];                        /* row-major, built once by the caller (Pinhole.cpp:109-112) */
  float ep[2];                         /* epipole in image 2 */
  const float* scale_factors2;         /* pKF2->mvScaleFactors */
  const float* level_sigma2_2;         /* pKF2->mvLevelSigma2 */
  int only_stereo, coarse, check_orientation;
} orc_tri_input;
/* ORBmatcher::SearchForTriangulation (ORBmatcher.cc:907-1146): fills matches12[n1] (-1 = none),
 * returns nmatches. */
int orc_search_triangulation(const orc_tri_input*, int* matches12);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)
 * (/root/reference/src/ORBmatcher.cc:223-425, single camera; Tracking::TrackReferenceKeyFrame, Tracking.cc:2798-2810).
 * Both sides in the orc_tri_input layout (kf = *1, frame = *2; has_mp1 = "map point present and not bad"; angle2 = F.mvKeys[].angle).
 * match2[i2] = index of the key-frame feature whose map point ends up in vpMapPointMatches[i2], or -1.  Returns nmatches. */
int orc_search_by_bow(const orc_tri_input* in, float nnratio, int check_orientation, int* match2);
/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (/root/reference/src/ORBmatcher.cc:765-905).
 * Both key frames in the orc_tri_input layout (has_mp = map point present and not bad, angle = mvKeysUn[].angle).
 * match12[i1] = index of the kf2 feature whose map point ends up in vpMatches12[i1], or -1.  Returns nmatches. */
int orc_search_by_bow_kf(const orc_tri_input* in, float nnratio, int check_orientation, int* match12);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (/root/reference/src/ORBmatcher.cc:1676-1887, single-camera case Nleft == -1) with
 * Frame::GetFeaturesInArea / AssignFeaturesToGrid / PosInGrid (src/Frame.cc:747-825, 475-506).  Flat arrays: */
typedef struct {
  int n1;                      /* LastFrame.N */
  const uint8_t* valid1;       /* mvpMapPoints[i] != NULL && !mvbOutlier[i] */
  const float* world_pos1;     /* pMP->GetWorldPos(), 3 floats per feature */
  const uint8_t* mp_desc1;     /* pMP->GetDescriptor(), 32 bytes per feature */
  const uint8_t* mp_observed1; /* pMP->Observations() > 0 (such a point blocks the feature it is assigned to) */
  const int32_t* octave1;      /* LastFrame.mvKeys[i].octave */
  const float* angle1;         /* LastFrame.mvKeysUn[i].angle */
  int n2;                      /* CurrentFrame.N */
  const float* kp2_xy;         /* CurrentFrame.mvKeysUn[i].pt */
  const int32_t* kp2_octave;
  const float* kp2_angle;
  const float* uright2;        /* CurrentFrame.mvuRight */
  const uint8_t* desc2;        /* CurrentFrame.mDescriptors */
  float grid[6];               /* mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv */
  float Tcw_q[4], Tcw_t[3];    /* CurrentFrame.GetPose(): unit quaternion (x, y, z, w) and translation */
  float Tlw_q[4], Tlw_t[3];    /* LastFrame.GetPose() */
  float K[4];                  /* fx, fy, cx, cy of CurrentFrame.mpCamera (Pinhole) */
  float mb, mbf;
  const float* scale_factors;  /* CurrentFrame.mvScaleFactors */
  int n_levels;
  float th;
  int mono;                    /* bMono */
  int check_orientation;       /* mbCheckOrientation */
} orc_projection_input;
/* match2[i2] = index of the LastFrame feature whose map point ends up in CurrentFrame.mvpMapPoints[i2], or -1.
 * Returns nmatches exactly as the reference counts it. */
int orc_search_by_projection(const orc_projection_input* in, int* match2);

/* The per-point search of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (/root/reference/src/ORBmatcher.cc:1340-1455,
 * from "Depth must be positive" to the threshold test) and of either direction of SearchBySim3 (:1497-1566 / :1575-1649), on
 * points that already are in the key frame's camera frame.  Same layout as rgbl_project_search_input. */
typedef struct {
  int n1;
  const uint8_t* valid1;
  const float* cam_pos1;
  const uint8_t* mp_desc1;
  const int32_t* level1;
  int n2;
  const float* kp2_xy;
  const int32_t* kp2_octave;
  const uint8_t* desc2;
  float grid[6];
  float K[4];
  const float* scale_factors;
  int n_levels;
  float th;
  int proj_form;   /* 0: Pinhole::project; 1: invz = 1.0 / z (double), u = fx * (x * invz) + cx */
  int max_dist;
} orc_project_search_input;
void orc_project_search(const orc_project_search_input* in, int* best_idx, int* best_dist);
/* ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming) (/root/reference/src/ORBmatcher.cc:427-532,
 * proj_form 0) and the vpPointsKFs overload (:534-646, proj_form 2: invz = 1 / z in float), from "Depth must be positive" on,
 * on camera-frame points; matched2 = vpMatched[i] != NULL on entry.  match2[i2] = point stored in vpMatched[i2] or -1.
 * Returns nmatches. */
int orc_search_by_projection_sim3(const orc_project_search_input* in, const uint8_t* matched2, int* match2);

/* MapPoint::ComputeDistinctiveDescriptors (/root/reference/src/MapPoint.cc:329-403) for a batch: descriptors of point p = rows
 * off[p] .. off[p+1]); best[p] = BestIdx (float distance table, std::sort of every row, median at (size_t)(0.5 * (N - 1)),
 * strict '<'), -1 for an empty list. */
void orc_distinctive_descriptors(const uint8_t* desc, const int32_t* off, int n_points, int32_t* best);

/* Frame::UndistortKeyPoints (/root/reference/src/Frame.cc:837-870): cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK).
 * OpenCV-internal arithmetic (cvUndistortPointsInternal with TermCriteria(MAX_ITER, 5, 0.01)), restated from upstream
 * knowledge - parity unpinned, like the other cv:: primitives.  K = fx, fy, cx, cy; dist = k1, k2, p1, p2[, k3]. */
void orc_undistort_points(const float* xy, int n, const float K[4], const float* dist, int n_dist, float* out_xy);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
 * (/root/reference/src/ORBmatcher.cc:1889-2010, single camera; Tracking::Relocalization) with MapPoint::PredictScale
 * (src/MapPoint.cc:531-546) and GetMin/MaxDistanceInvariance (src/MapPoint.cc:502-512).  Flat arrays, the key frame's map
 * points as the reference sees them: */
typedef struct {
  int n1;                      /* pKF->GetMapPointMatches().size() */
  const uint8_t* has_mp1;      /* vpMPs[i] != NULL */
  const uint8_t* bad1;         /* pMP->isBad() */
  const uint8_t* found1;       /* sAlreadyFound.count(pMP) */
  const float* world_pos1;     /* pMP->GetWorldPos() */
  const uint8_t* mp_desc1;     /* pMP->GetDescriptor() */
  const float* min_dist1;      /* pMP->mfMinDistance (GetMinDistanceInvariance = 0.8f * it) */
  const float* max_dist1;      /* pMP->mfMaxDistance (GetMaxDistanceInvariance = 1.2f * it; PredictScale uses it raw) */
  const float* angle1;         /* pKF->mvKeysUn[i].angle */
  int n2;
  const float* kp2_xy;         /* CurrentFrame.mvKeysUn[i].pt */
  const int32_t* kp2_octave;
  const float* kp2_angle;
  const uint8_t* desc2;
  const uint8_t* occupied2;    /* CurrentFrame.mvpMapPoints[i] != NULL on entry */
  float grid[6];
  float Tcw_q[4], Tcw_t[3];
  float K[4];
  const float* scale_factors;
  int n_levels;
  float log_scale_factor;      /* CurrentFrame.mfLogScaleFactor */
  float th;
  int orb_dist;
  int check_orientation;
} orc_kf_projection_input;
/* match2[i2] = index of the key-frame feature whose map point the call stores in CurrentFrame.mvpMapPoints[i2], or -1 (left
 * as it was).  Returns nmatches. */
int orc_search_by_projection_kf(const orc_kf_projection_input* in, int* match2);
/* What the caller of the C ABI (the shim) evaluates with the MapPoint objects: valid1[i] (good, not found yet, distance to
 * the camera centre inside the invariance range) and level1[i] = PredictScale. */
void orc_kf_projection_prepass(const orc_kf_projection_input* in, uint8_t* valid1, int32_t* level1);

/* The per-point search of ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th, bRight = false)
 * (/root/reference/src/ORBmatcher.cc:1148-1338, single camera) with KeyFrame::GetFeaturesInArea / IsInImage
 * (src/KeyFrame.cc:704-753) and MapPoint::PredictScale(dist, KeyFrame*) (src/MapPoint.cc:514-529). */
typedef struct {
  int n1;                      /* vpMapPoints.size() */
  const uint8_t* has_mp1;      /* vpMapPoints[i] != NULL */
  const uint8_t* bad1;         /* pMP->isBad() */
  const uint8_t* in_kf1;       /* pMP->IsInKeyFrame(pKF) */
  const float* world_pos1;
  const float* normal1;        /* pMP->GetNormal() */
  const uint8_t* mp_desc1;
  const float* min_dist1;      /* mfMinDistance */
  const float* max_dist1;      /* mfMaxDistance */
  int n2;
  const float* kp2_xy;         /* pKF->mvKeysUn[i].pt */
  const int32_t* kp2_octave;
  const float* uright2;
  const uint8_t* desc2;
  float grid[6];
  float Tcw_q[4], Tcw_t[3];
  float Ow[3];                 /* pKF->GetCameraCenter() */
  float K[4];
  float bf;
  const float* scale_factors;
  const float* inv_level_sigma2;
  int n_levels;
  float log_scale_factor;      /* pKF->mfLogScaleFactor */
  float th;
} orc_fuse_input;
/* best_idx[i] = bestIdx of point i when bestDist <= TH_LOW, else -1 (also for skipped points).  Returns nFused. */
int orc_fuse_search(const orc_fuse_input* in, int* best_idx);
/* valid1 / level1 as the caller of rgbl_fuse_search has to provide them. */
void orc_fuse_prepass(const orc_fuse_input* in, uint8_t* valid1, int32_t* level1);

/* ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)
 * (/root/reference/src/ORBmatcher.cc:43-213, single camera; called by Tracking::SearchLocalPoints, Tracking.cc:3447):
 * local map points that Frame::isInFrustum found visible are searched around their predicted projection. */
typedef struct {
  int n1;                      /* vpMapPoints.size() */
  const uint8_t* valid1;       /* mbTrackInView && !(bFarPoints && mTrackDepth > thFarPoints) && !isBad() */
  const float* proj1;          /* mTrackProjX, mTrackProjY, mTrackProjXR: 3 floats per point */
  const int32_t* level1;       /* mnTrackScaleLevel */
  const float* view_cos1;      /* mTrackViewCos */
  const uint8_t* mp_desc1;     /* GetDescriptor(), 32 bytes per point */
  const uint8_t* mp_observed1; /* Observations() > 0 */
  int n2;                      /* F.N */
  const float* kp2_xy;         /* F.mvKeysUn[i].pt */
  const int32_t* kp2_octave;
  const float* uright2;        /* F.mvuRight */
  const uint8_t* desc2;        /* F.mDescriptors */
  const uint8_t* blocked2;     /* F.mvpMapPoints[i] != NULL && ->Observations() > 0 on entry */
  float grid[6];               /* mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv */
  const float* scale_factors;  /* F.mvScaleFactors */
  int n_levels;
  float th;
  float nnratio;               /* mfNNratio */
} orc_local_points_input;
/* match2[i2] = index of the map point the call assigns to F.mvpMapPoints[i2], or -1 (left as it was). Returns nmatches. */
int orc_search_local_points(const orc_local_points_input* in, int* match2);
/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
 * (/root/reference/src/ORBmatcher.cc:648-763) with Frame::GetFeaturesInArea (src/Frame.cc:747-813). */
typedef struct {
  int n1;
  const int32_t* kp1_octave;
  const float* kp1_angle;
  const uint8_t* desc1;
  int n2;
  const float* kp2_xy;
  const int32_t* kp2_octave;
  const float* kp2_angle;
  const uint8_t* desc2;
  float grid[6];
  int window_size;
  float nnratio;
  int check_orientation;
} orc_initialization_input;
/* prev_matched: in / out (2 floats per F1 feature); matches12: n1 entries. Returns nmatches. */
int orc_search_for_initialization(const orc_initialization_input* in, float* prev_matched, int* matches12);
/* F12 = K1^-T [t]x R12 K2^-1 with Eigen's evaluation order in fp32 */
void orc_fundamental(const float K1[4], const float K2[4], const float R12[9], const float t12[3],
                     float F12[9]);

#ifdef __cplusplus
}
#endif
#endif
