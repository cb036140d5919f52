/*
 * ba_oracle.h — CPU restatement of the reference bundle-adjustment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (the HIP library under
 * cuda-bundle-adjustment_amd/) may include, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker / the
 * timed CPU baseline — never as the thing shipped.
 *
 * PARITY UNPINNED: the reference ships no tests, no golden vectors, its datasets
 * (samples/ba_input.7z) are absent, and it cannot be built here (needs nvcc, cuSOLVER,
 * cuSPARSE, thrust, Eigen).  This restatement is therefore pinned only by (a) an
 * independent numpy restatement (tests/golden/make_golden.py) of the same formulas,
 * (b) finite-difference checks of the Jacobians, and (c) the README chi2 table kept as a
 * dormant known-answer test that activates if ba_kitti_00.json is ever supplied.
 *
 * All file:line citations are into /root/reference.
 * Layouts follow the reference: quaternion (x,y,z,w); pose = world->camera; small
 * matrices column-major (src/cuda/cuda_block_solver.cu:64-70).
 */
#ifndef BA_ORACLE_H
#define BA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { BA_RK_NONE = 0, BA_RK_CAUCHY = 1, BA_RK_TUKEY = 2, /* src/robust_kernel.h:12-17 */
       BA_RK_HUBER = 3 /* extension beyond the reference: g2o RobustKernelHuber */ };

/* A BA problem as flat arrays.  Vertices are listed in ascending-id order (the order
 * std::map iteration gives the reference, src/optimisable_graph.hpp:95).  Edges refer to
 * vertices by position in these arrays. */
typedef struct ba_problem
{
    int n_poses, n_landmarks, n_edges;
    double* pose;              /* n_poses x 7: qx qy qz qw tx ty tz                       */
    unsigned char* pose_fixed; /* n_poses                                                  */
    double* lm;                /* n_landmarks x 3                                          */
    unsigned char* lm_fixed;   /* n_landmarks                                              */
    int* e_pose;               /* n_edges: position into pose[]                            */
    int* e_lm;                 /* n_edges: position into lm[]                              */
    unsigned char* e_stereo;   /* n_edges: 0 = MonoEdge (dim 2), 1 = StereoEdge (dim 3)    */
    double* e_meas;            /* n_edges x 3 (mono uses the first two)                    */
    double* e_omega;           /* n_edges: scalar information                              */
    double* e_cam;             /* n_edges x 5: fx fy cx cy bf                              */
    int rk_type;               /* BA_RK_*; one kernel for all edges (reference: global)    */
    double rk_delta;
} ba_problem;

/* Per-LM-iteration trace, mirrors what optimize() prints (cuda_graph_optimisation.cpp:113-131) */
typedef struct ba_iter_info
{
    int iteration;
    double chi2;   /* F recorded by stats_.addStat              */
    double lambda; /* lambda after the iteration                */
    double rho;    /* last rho                                  */
    int trials;    /* q: number of rejected trials              */
} ba_iter_info;

/* Single-edge evaluation (cuda_block_solver.cu:379-424, 449-578, 1060-1110, 1152-1220).
 * dim = 2 (mono) or 3 (stereo).  JP is dim x 6, JL is dim x 3, both column-major.
 * w = omega * rho'(omega*|e|^2). Any output pointer may be NULL. */
void ba_edge_eval(const double* pose7, const double* Xw3, const double* meas, int dim,
                  double omega, const double* cam5, int rk_type, double rk_delta,
                  double* e, double* Xc, double* chi, double* JP, double* JL, double* w);

/* exp-map pose update T <- exp([w,v]) * T  (cuda_block_solver.cu:781-823, 1444-1455) */
void ba_pose_update(double* pose7, const double* dx6);

/* symmetric 3x3 inverse by adjugate (cuda_block_solver.cu:639-669); A,B column-major 3x3 */
void ba_sym3_inv(const double* A, double* B);

/* robust kernel value / derivative (cuda_block_solver.cu:972-1027) */
double ba_rk_rho(int type, double delta, double x);
double ba_rk_drho(int type, double delta, double x);

/* Index assignment: free vertices first (ascending id), fixed after
 * (optimisable_graph.hpp:84-126). idx arrays have n_poses / n_landmarks entries.
 * Returns number of free poses in *np_free and free landmarks in *nl_free. */
void ba_assign_indices(const ba_problem* p, int* pose_idx, int* lm_idx, int* np_free,
                       int* nl_free);

/* total chi2 at the current estimates (block_solver.cpp:250-272): sum over the mono set
 * then the stereo set. Optionally writes per-edge errors (n_edges x 3), Xc (n_edges x 3). */
double ba_compute_errors(const ba_problem* p, double* errors, double* Xcs);

/* Normal equations at the current estimates (block_solver.cpp:274-307). Output blocks are
 * indexed by the free-vertex indices of ba_assign_indices:
 *   Hpp: np_free x 36 (6x6 col-major), bp: np_free x 6, Hll: nl_free x 9, bl: nl_free x 3,
 *   Hpl: n_edges x 18 (6x3 col-major; zero for edges with a fixed vertex).
 * Any output may be NULL. Returns chi2. */
double ba_build_system(const ba_problem* p, double* Hpp, double* bp, double* Hll, double* bl,
                       double* Hpl);

/* Damped Schur complement as a dense matrix (block_solver.cpp:322-388,
 * cuda_block_solver.cu:1286-1345): Hsc is (6 np_free)^2 row-major==col-major (symmetric,
 * full), bsc is 6 np_free. */
void ba_schur_dense(const ba_problem* p, double lambda, double* Hsc, double* bsc);

/* One damped solve at the current estimates: returns 1 on success, 0 on zero pivot.
 * dx_p: np_free x 6, dx_l: nl_free x 3.  use_dense!=0 forces the dense LL^T. */
int ba_solve_step(const ba_problem* p, double lambda, int use_dense, double* dx_p,
                  double* dx_l);

/* Full Levenberg-Marquardt run (cuda_graph_optimisation.cpp:48-154). Estimates in p are
 * updated in place.  info[] must have room for niterations entries; returns the number of
 * iterations recorded.  use_dense: 0 = block-sparse LL^T with minimum-degree ordering,
 * 1 = dense LL^T (small problems; independent check of the sparse solver). */
int ba_optimize(ba_problem* p, int niterations, int use_dense, ba_iter_info* info);

/* ---- sparse SPD block solver exposed for cross-checks ------------------------------- */
/* Solve A x = b where A is given as upper-triangular block CSR (6x6 blocks col-major,
 * columns ascending, diagonal first in each row). Returns 1 ok / 0 zero pivot. */
int ba_bsr_chol_solve(int nb, const int* rowptr, const int* colind, const double* vals,
                      const double* b, double* x);

#ifdef __cplusplus
}
#endif
#endif
