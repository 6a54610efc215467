/* TEST INFRASTRUCTURE ONLY.  Float64 CPU restatement of the reference's centroidal-MPC NLP and a
 * reference interior-point solver for it.  Nothing in the shipped package links or loads this;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * The NLP restated here is the one the reference ships as CasADi-generated C:
 *   /root/reference/src/centroidal-mpc-walking/config/robots/ergoCubGazeboV1/tmp.c
 *     nlp_fg :12430   nlp_grad :24791   nlp_hess_l :58926   nlp_jac_fg :71962   sparsity :62-67
 * (generated for N=12, dt=0.1; here N, dt, weights, corners and mu are runtime parameters).
 * Pinned in tests/test_oracle_nlp.py against oracle/_ref (that generated code compiled as is)
 * and against tests/golden/nlp_*.npz produced from it by tests/golden/make_nlp_golden.py.
 */
#ifndef CMPC_ORACLE_H
#define CMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define CMPC_NC 2     /* contacts: left_foot, right_foot (std::map order) */
#define CMPC_NCORN 4  /* corners per contact */

typedef struct {
    int N;                 /* horizon (number of control intervals) */
    double dt;             /* sampling_time */
    double mu;             /* static_friction_coefficient */
    double gravity;        /* 9.80665 */
    double w_com[3];       /* com_weight */
    double w_h;            /* angular_momentum_weight */
    double w_pos;          /* contact_position_weight */
    double w_rate[3];      /* force_rate_of_change_weight */
    double w_sym;          /* contact_force_symmetry_weight */
    double corners[CMPC_NC][CMPC_NCORN][3]; /* corner_j, foot frame */
} cmpc_nlp_cfg;

/* sizes: n_x = 45N+15, n_p = 50N+27, n_g = 53N+15, nnz J = 243N+15, nnz H = 348N-36 */
void cmpc_nlp_dims(const cmpc_nlp_cfg* c, int* nx, int* np, int* ng, int* nnzj, int* nnzh);

/* f(x,p), g(x,p) */
void cmpc_nlp_fg(const cmpc_nlp_cfg* c, const double* x, const double* p, double* f, double* g);
/* grad_x f */
void cmpc_nlp_grad_f(const cmpc_nlp_cfg* c, const double* x, const double* p, double* gf);
/* jac_x g as COO triplets (structural pattern incl. entries that are numerically 0); returns nnz */
int cmpc_nlp_jac(const cmpc_nlp_cfg* c, const double* x, const double* p, int* row, int* col,
                 double* val);
/* hess_xx (lam_f f + lam_g^T g), full symmetric, COO; returns nnz */
int cmpc_nlp_hess(const cmpc_nlp_cfg* c, const double* x, const double* p, double lam_f,
                  const double* lam_g, int* row, int* col, double* val);

/* nlp_grad: gradient of lam_f f + lam_g^T g w.r.t. x (gx[nx]) and p (gp[np]); either may be NULL */
void cmpc_nlp_grad(const cmpc_nlp_cfg* c, const double* x, const double* p, double lam_f, const double* lam_g,
                   double* gx, double* gp);

/* ---- reference interior-point solver (ipm_ref.c): see that file's header ---- */
typedef struct {
    int max_iter;       /* Newton iteration budget */
    double tol;         /* KKT tolerance (scaled inf-norm, like ipopt_tolerance) */
    double mu_init;     /* initial barrier parameter */
    double mu_min;      /* smallest barrier parameter */
    int exact_hessian;  /* 1: Lagrangian Hessian incl. the bilinear momentum term; 0: Gauss-Newton */
    int verbose;
    int tail_stages;    /* > 0: re-solve the last tail_stages stages after convergence when the extrapolation step there is large */
    int tail_iters;     /* Newton steps of that re-solve (before its own affine-scaling step) */
    double tail_trigger;/* ... larger than tail_trigger x the largest force component */
} cmpc_ipm_opts;

#ifdef __cplusplus
}
#endif
#endif
