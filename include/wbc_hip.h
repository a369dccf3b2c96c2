/* wbc_hip.h -- C-ABI of the MI355X (gfx950) batched whole-body-control hot path.
 *
 * Drop-in boundary for the per-tick computation of `dogbot_controller`
 * (reference: /root/reference/.gitmodules:4-6; the controller is started as
 * `rosrun dogbot_controller dogbot <path>/dogbot.urdf`, /root/reference/README.md:60, and
 * contains "a momentum-based observer estimator, and an optimization problem based on the
 * modulation of ground reaction forces", /root/reference/README.md:11).
 *
 * PROVISIONAL: the controller's source is an un-vendored submodule that is absent from
 * /root/reference, so the names and signatures of its compute-torques entry point and its ROS
 * message types cannot be cited.  Each entry point below states which reference
 * interface it stands for; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Rules of the ABI: extern "C"; plain pointers and sizes; no exceptions cross it; every function
 * returns an int status (WBC_OK = 0); the caller owns every batch buffer; a solver is bound to one
 * HIP device, is not thread-safe, distinct solvers are; no allocation or synchronisation happens
 * inside wbc_step_batch / wbc_dynamics_batch (they are hipGraph-capturable -- except while per-kernel
 * timing is enabled: a dispatch that carries timing events cannot be captured).  Every entry point runs
 * on its solver's device and restores the caller's current HIP device before it returns.
 *
 * Batch layout in HBM: structure-of-arrays, component-major: x[c * N + s] is component c of state s
 * (N = batch size of the call).  Scalars are double (WBC_F64) or float (WBC_F32) per the solver.
 *   q        [nq = 7+nj][N]  base position (world), base quaternion (x,y,z,w), joint angles
 *   v        [nv = 6+nj][N]  base linear velocity (world), base angular velocity (world), joint rates
 *   w_des    [6][N]          desired contact wrench on the base rows (force; moment about base origin)
 *   vdot_des [nv][N]         desired generalized acceleration
 *   normals  [3*nf][N]       terrain normal under each foot (world)
 *   mu       [nf][N]         friction coefficient under each foot
 *   mask     [N] int32       bit k set = foot k in stance
 *   tau_prev [nj][N], f_prev [3*nf][N]   previous tick's outputs (observer only)
 *   obs_integ, obs_r [nv][N] observer state, in/out (observer only)
 *   tau [nj][N], f [3*nf][N], status [N] int32 (0 ok, 1 QP iteration limit, 2 QP infeasible), iters [N] int32
 *   M [nv(nv+1)/2][N]  packed upper triangle, idx(i,j) = i*nv - i(i-1)/2 + (j-i), i <= j
 *   h [nv][N], Jc [3*nf*nv][N] (row (3k+m)*nv + c), pf [3*nf][N]
 */
#ifndef WBC_HIP_H
#define WBC_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WBC_MAXV 32 /* capacity of the per-dof gain arrays */

enum wbc_status {
  WBC_OK = 0,
  WBC_E_INVALID = 1,  /* null pointer / bad argument */
  WBC_E_IO = 2,       /* URDF file cannot be read */
  WBC_E_PARSE = 3,    /* URDF malformed or uses an unsupported element */
  WBC_E_TOPOLOGY = 4, /* not a floating base with 4 legs x 3 revolute joints + 4 feet */
  WBC_E_NODEVICE = 5, /* no HIP device / kernels for gfx950 cannot run here */
  WBC_E_HIP = 6,      /* a HIP runtime call failed (see wbc_last_error) */
  WBC_E_CAPACITY = 7  /* N exceeds the solver's max_batch */
};

enum wbc_dtype { WBC_F64 = 0, WBC_F32 = 1 };

typedef struct wbc_model wbc_model;   /* host-side flattened robot model */
typedef struct wbc_solver wbc_solver; /* device-side context */

/* Controller parameters.  In the reference these are presumably compile-time constants of the
 * controller ([UNVERIFIED]); here all are run-time. */
typedef struct wbc_params {
  double S[6];        /* wrench-tracking weights: force xyz, moment xyz */
  double alpha;       /* GRF regularisation (> 0) */
  double fn_min;      /* normal-force bounds per stance foot */
  double fn_max;
  double mu_scale;    /* 1 = pyramid with mu, 1/sqrt(2) = inscribed pyramid */
  double dt;          /* control period, s */
  int observer_order; /* 0 = off, 1, 2 */
  int max_iter;       /* QP active-set iteration cap */
  double qp_tol;      /* constraint-violation tolerance, N */
  double K1[WBC_MAXV];
  double K2[WBC_MAXV];
} wbc_params;

/* ---- model: stands for the controller's URDF ingestion (argv[1], README.md:60) ---- */
/* foot_links may be NULL: then the last link hanging off each leg's distal body is the foot. */
int wbc_model_load_urdf(const char* path, const char* const* foot_links, int n_foot_links, wbc_model** out);
int wbc_model_from_flat(int nb, const int* parent, const double* Rt, const double* rt, const double* axis,
                        const double* mass, const double* com, const double* Ic, int nf, const int* foot_body,
                        const double* foot_off, const double* gravity, wbc_model** out);
void wbc_model_free(wbc_model* m);
int wbc_model_dims(const wbc_model* m, int* nb, int* nq, int* nv, int* nj, int* nf);
/* copies the flat arrays out (sizes per wbc_model_dims; any pointer may be NULL) */
int wbc_model_get_flat(const wbc_model* m, int* parent, double* Rt, double* rt, double* axis, double* mass,
                       double* com, double* Ic, int* foot_body, double* foot_off, double* gravity);
const char* wbc_model_joint_name(const wbc_model* m, int j); /* NULL if out of range / unnamed */
const char* wbc_model_foot_link(const wbc_model* m, int k);
double wbc_model_total_mass(const wbc_model* m);

void wbc_params_default(wbc_params* p, int dtype);

/* ---- solver ---- */
int wbc_solver_create(const wbc_model* m, const wbc_params* p, int dtype, int device, size_t max_batch,
                      wbc_solver** out);

/* Kernel-selection options.  The defaults are the measured winners; nothing in the library reads the
 * environment, so a C++ host sees every switch here.  Call wbc_solver_options_default first, then change fields. */
enum wbc_timing_mode { WBC_TIMING_DISPATCH = 0, /* the dispatch's own start/stop timestamps (what rocprofv3 reports) */
                       WBC_TIMING_EVENT_PAIR = 1 /* an event pair recorded around the launch (+2-3 us per span) */ };
typedef struct wbc_solver_options {
  size_t struct_size;     /* sizeof(wbc_solver_options) of the caller's build (set by wbc_solver_options_default) */
  long long fused_max;    /* ticks of at most this many states run as ONE launch of wavefront roles; -1 = auto
                             (11264, observer on 12288; rollouts: 4096 -- wbc_plan_tick / wbc_dispatch_thresholds report it; at auto, fp64 observer-off ticks with matrix outputs
                             leave the one-launch tick at 8192 states already: tile_tick below), 0 = always the two-kernel tick */
  int rollout_persistent; /* 1 (default): rollouts of at most fused_max (auto: 4096) states = one launch per rollout; 0: per-tick launches */
  int rollout_spw;        /* states per workgroup of the rollout kernel: 0 = auto (4 up to 1024 states, else 16), 4, 16 */
  long long obs_split_min;/* observer-on two-kernel ticks of at least this many states run the observer update as its own
                             kernel instead of inside the sweep; -1 = auto (fp32: from 33792 states on, fp64: from 20480),
                             -2 = never */
  int one_zerocopy;       /* (default 3) single-robot host-pointer calls: 0 = staging copies + hipStreamSynchronize; 1 = the kernel reads / writes the
                             pinned image directly (mapped host memory); 2 = as 1, and completion is a ticket the stream writes into the
                             image behind the tick (hipStreamWriteValue32), polled by the host in memory: no runtime call on the wait
                             path; 3 = as 2 with a one-thread kernel writing the ticket.  2 / 3 fall back to 1 when the stack refuses */
  int timing_mode;        /* enum wbc_timing_mode, used by wbc_solver_enable_timing */
  int qp_tile;            /* GRF-QP kernel of the two-kernel tick: 0 = auto (tiles of states dealt to the wavefronts by predicted work,
                             sized so that the launch is ONE round of resident workgroups: fp64 from 14336 states on, 32 ... 64 states
                             per tile; fp32 from 16384 to 49152 states STAGED tiles -- one workgroup per CU holds ceil(N / 256) states
                             and their inputs in LDS -- then 72 ... 128 states per tile; one-wavefront workgroups below), -1 = never
                             tiles, otherwise always tiles of that many states: 32 | 64 | 128 | 256 | 512, fp64 also 36 ... 60 in
                             steps of 4, fp32 also every multiple of 4 up to 192 (staged) */
  int obs_split_serial;   /* that observer kernel runs 1 (default) = on the caller's stream before the sweep, 0 = beside it on a
                             second stream (measured slower: the two compete for the same SIMDs) */
  int qp_lane;            /* two-kernel ticks solve the QPs one state per LANE first (semismooth Newton on the residual wrench)
                             and send what that does not finish to the dense active-set kernel: 0 = auto (fp64 batches from
                             106496 states on, fp32 from 212992), 1 = always, -1 = never.  States solved per lane report status 0 and
                             iters = Newton iterations (<= 5); the others the dense kernel's status / iteration count.
                             wbc_params.qp_tol and max_iter govern the DENSE kernel only: the per-lane kernel accepts a state when
                             the residual of its optimality equation is below 1e-11 (fp32: 2e-5) x (1 + |target wrench|) or a full
                             Newton step leaves the active faces unchanged (then the point is the exact solution for those faces) */
  int f32_pack2;          /* fp32 dynamics sweep with TWO states per lane (packed v_pk_* arithmetic, whole 128-byte lines per 16-lane
                             row, half the wavefronts): 0 = auto (even batches from 32768 states on), 1 = every even batch, -1 = never */
  int keep_structural;    /* 54 of the 171 packed-M words and 108 of the 216 Jc words per state are structural zeros / ones (cross-leg
                             blocks, identity and skew-diagonal entries): 37 % of what the dynamics sweep stores.  1: a call that gets the
                             SAME M and Jc buffers and the same N as the previous call on this solver does not rewrite them -- the caller
                             promises not to touch M / Jc between ticks (any other pointer or N: written in full again).  0 (default):
                             every call writes every word.  The buffers are identified by ADDRESS: freeing and reallocating them (or a
                             caching allocator handing the same address out again) counts as touching them -- call
                             wbc_solver_invalidate_structural then */
  int rollout_warm;       /* (default 1) wbc_rollout_batch / wbc_rollout_tracking_batch: every tick after the first starts its GRF QPs from the
                             active set of the previous tick (see wbc_step_batch_warm).  The QP is strictly convex, so the results do not
                             depend on it -- the time per tick does.  0: every tick solves from the unconstrained minimum */
  int multi_threads;      /* (ABI 7; read by wbc_multi_create only) one persistent ISSUE THREAD per shard, bound to the shard's device: an entry point
                             validates every shard on the caller's thread, posts one ticket, the threads enqueue their shards in parallel and the call
                             returns when all have -- eight devices are no longer fed one launch after the other from one thread.  1 = on, -1 = never (the
                             serial issue of ABI <= 6), 0 = auto = serial (ABI 8: the threads are opt-in until they have been run on more than one device) */
  int multi_spin_us;      /* (default 200) an idle issue thread polls for its next ticket this long before it parks on a condition variable: a tick loop
                             never pays a wake-up, a 1 kHz control loop does not burn a core per shard */
  int obs_colaunch;       /* (ABI 7) observer-on two-kernel ticks with M/h/Jc outputs: the observer update and the observer-free sweep as the TWO ROLES OF
                             ONE LAUNCH (wbc_tick_plan.front = 4) while both roles' wavefronts are resident together -- fp32 batches of 12290 ... 32768
                             states (even; both roles with two states per lane), BASELINE's configs[3] shard; fp64 batches of 12289 ... 14336.
                             0 = auto, 1 = whenever the tick is a two-kernel tick with the observer on, -1 = never */
  int tile_tick;          /* (ABI 8) fp32 observer-on ticks with M/h/Jc outputs of an even batch: ONE launch of 128-state workgroups, one per CU -- the sweep and
                             observer roles of obs_colaunch side by side in a workgroup, then the staged QP tile of the same states behind one barrier
                             (wbc_tick_plan.fused = 2).  0 = auto (from 8194 states on -- up to 12288 in front of the one-launch tick while fused_max is at auto; 64 / 96 / 128 states per workgroup: one round of workgroups up to 32768 states,
                             BASELINE's configs[3] shard), 1 = every such tick beyond the fused_tick size, -1 = never.  Also fp64 observer-off ticks of 8193 ... 28672 states (48 ... 112-state workgroups, ahead of the one-launch tick while
                             fused_max is at auto; 1: every size beyond the fused_tick size) and fp64 observer-on ticks of 8193 ... 196608 states (32 / 48 / 64-state
                             workgroups of sweep + observer wavefronts, 64-state ones in rounds beyond 16384 states).  fp32 observer-off ticks: only with tile_tick = 1
                             (measured: +16 % at 32768 states, a loss at 49152 and below 16384 -- profiles/r06q_tile_tick_f32_noobs.log).  Auto applies only while qp_tile, qp_lane, obs_colaunch and obs_split_min are at auto themselves */
  int fused_pair;         /* (ABI 9) observer-off cold ticks with M/h/Jc outputs of N >= 64 states: the one-launch tick as twelve-wavefront workgroups of 32 states
                             (wbc_tick_plan.fused = 3: two of the 16-state workgroups in one, at 168 registers, so that BOTH halves are resident on a CU together -- 8192 states are
                             one round of workgroups instead of two; a batch that is not a multiple of 32 gets one more workgroup anchored at its end).  0 = auto (fp64 4225 ... 8192 states, fp32 4225 ... 16384, wbc_dispatch_thresholds reports them; only while
                             fused_max, tile_tick and the kernel selectors above are at auto), 1 = every such tick up to 65536 states, -1 = never */
} wbc_solver_options;
void wbc_solver_options_default(wbc_solver_options* o);
int wbc_solver_create_ex(const wbc_model* m, const wbc_params* p, int dtype, int device, size_t max_batch,
                         const wbc_solver_options* opt /* NULL = defaults */, wbc_solver** out);
int wbc_solver_device(const wbc_solver* s); /* HIP device index, -1 for NULL */
/* keep_structural: the next call writes M / Jc in full again whatever buffers it gets */
int wbc_solver_invalidate_structural(wbc_solver* s);

/* Which kernels a tick of N states runs with these options (no device needed): the measured switches of DESIGN.md section 5 as data, so
 * that a caller -- and the parity tests, which straddle every switch -- never restate them.  All fields are informational. */
typedef struct wbc_tick_plan {
  size_t struct_size; /* in: sizeof of the caller's build (0 = this build's); out: bytes written */
  int fused;          /* 1: the whole tick is ONE fused_tick launch (wavefront roles); everything below is 0 then.  3 (ABI 9): the same as 32-state workgroups
                         (fused_pair_kernel, wbc_solver_options.fused_pair).  2: ONE tile_tick launch (sweep | observer roles,
                         then the staged QP tile of the same states: front = 4, qp = 1, qp_body = 2, qp_tile = states per workgroup say what runs inside it) */
  int front;          /* two-kernel ticks, front half: 0 = dyn_sweep (observer inside when on), 1 = rnea_step (caller passes no M/h/Jc),
                         2 = observer kernel + observer-free dyn_sweep, 3 = observer kernel + observer-free rnea_step (no M/h/Jc),
                         4 = observer update and observer-free dyn_sweep as the two roles of ONE launch (sweep_obs_kernel) */
  int qp;             /* 0 = qp_group16 (one-wavefront workgroups), 1 = qp_tile (tiles dealt by predicted work), 2 = qp_lane + qp_list */
  int qp_tile;        /* states per tile when qp == 1 */
  int qp_body;        /* 0 = wrench-space dual active set in fp64 arithmetic, 1 = 12 x 12 orthogonal-factor body in fp32 (fp32 tiles beyond 65 536 states),
                         2 = as 0 on STAGED tiles: the tile's inputs go through LDS, results are stored row by row (fp32 solvers, tiles <= 192) */
  int sweep_pack2;    /* fp32 dyn_sweep with two states per lane */
  int sweep_block;    /* threads per workgroup of the dyn_sweep launch (64 / 256); 0 when no dyn_sweep runs */
  int qp_warm;        /* wbc_step_batch_warm: 1 = the QP kernels START from the carried active sets, 0 = they only report them (the sizes at which
                         the cold tiles are the faster kernels) */
} wbc_tick_plan;
/* warm != 0: the plan of wbc_step_batch_warm */
int wbc_plan_tick(int dtype, int observer_order, const wbc_solver_options* opt /* NULL = defaults */, size_t N, int with_mats, int with_pf,
                  int warm, wbc_tick_plan* plan);
int wbc_solver_plan_tick(const wbc_solver* s, size_t N, int with_mats, int with_pf, int warm, wbc_tick_plan* plan);
/* the batch sizes N at which the plan differs from that of N - 1 (fp32: N - 2; odd batches never pack), ascending; *n = how many
 * (at most 16), the first min(*n, cap) are written to out.  flags: WBC_PLAN_WITH_MATS (the caller passes M / h / Jc buffers) |
 * WBC_PLAN_WARM (the switches of wbc_step_batch_warm) */
#define WBC_PLAN_WITH_MATS 1
#define WBC_PLAN_WARM 2
int wbc_dispatch_thresholds(int dtype, int observer_order, const wbc_solver_options* opt, int flags, size_t* out, int cap, int* n);
/* diagnostics (synchronises the device): states of the last two-kernel tick that the per-lane QP kernel handed to the dense one */
int wbc_solver_qp_handover(wbc_solver* s, int* count);
void wbc_solver_destroy(wbc_solver* s);
int wbc_solver_set_params(wbc_solver* s, const wbc_params* p);

typedef struct wbc_batch_in {
  const void* q;
  const void* v;
  const void* w_des;
  const void* vdot_des;
  const void* normals;
  const void* mu;
  const int* mask;
  const void* tau_prev; /* observer only, else may be NULL */
  const void* f_prev;   /* observer only, else may be NULL */
} wbc_batch_in;

typedef struct wbc_batch_out {
  void* tau;
  void* f;
  int* status;
  int* iters; /* may be NULL.  The solver's own count per state -- dual active-set trips (one-launch ticks with the observer on: of both phases of the
               * speculative start), Newton steps where the per-lane kernel solved the state; informational, never part of the parity contract */
  void* M;    /* optional dynamics outputs: all NULL or M, h, Jc all non-NULL */
  void* h;
  void* Jc;
  void* pf;   /* may be NULL */
} wbc_batch_out;

/* Momentum-observer state, in/out across ticks.  START-UP CONTRACT: the residual is r = K1 (M v - integ), so before the
 * first observer-on tick integ must hold the generalized momentum p(0) = M(q0) v0 of the state the loop starts from and
 * r must be zero -- zeros in integ are only right for a robot at rest (otherwise the first ticks carry a spurious
 * estimate of about K1 p(0) that decays with time constant 1/K1).  Batch callers get p from
 * wbc_dynamics_batch(..., p = integ, ...); the single-robot loop calls wbc_observer_init. */
typedef struct wbc_observer_state {
  void* integ;
  void* r;
} wbc_observer_state;

/* Dynamics sweep only: M(q), h(q,v), Jc(q) (+ optional pf, p = M v, beta = C^T v - g).
 * Stands for the rigid-body-dynamics calls the controller makes each tick (north_star:
 * "CRBA mass matrix, RNEA bias forces, contact Jacobians from the DogBot URDF").
 * All pointers are device pointers on the solver's device; `stream` is a hipStream_t (NULL = default). */
int wbc_dynamics_batch(wbc_solver* s, size_t N, const void* q, const void* v, void* M, void* h, void* Jc,
                       void* pf, void* p, void* beta, void* stream);

/* One control tick for N independent states: dynamics -> observer -> GRF QP -> torque map.
 * Stands for the controller's per-tick compute-torques entry point (name [UNVERIFIED]). */
int wbc_step_batch(wbc_solver* s, size_t N, const wbc_batch_in* in, const wbc_batch_out* out,
                   const wbc_observer_state* obs, void* stream);

/* One tick of a DEPENDENT sequence -- the reference controller runs in a loop on one robot (/root/reference/README.md:58-62), and
 * consecutive ticks share most of the active set of their GRF QP.  active_in [N] int32 (may be NULL = start cold) is the active set
 * each state's dual active-set iteration starts from: normally what the previous tick wrote to active_out [N] (may be NULL; in
 * place allowed).  Encoding, per state: for foot k with unit normal n, tangents t1, t2 and mu~ = mu * mu_scale
 *     bit 4k + j,      j = 0 .. 3:  the rows  (mu~ n - t1) . f >= 0,  (mu~ n - t2) . f >= 0,  n . f >= fn_min,  -n . f >= -fn_max
 *     bit 16 + 4k + j, j = 0 .. 1:  the rows  (mu~ n + t1) . f >= 0,  (mu~ n + t2) . f >= 0
 * Bits of swing feet are ignored.  The set is a HINT: the solver builds the minimiser on it in one block step and continues the
 * iteration from there when it is an S-pair of the dual method (independent rows, non-negative multipliers), otherwise it starts
 * cold -- the QP is strictly convex, so tau, f and status never depend on the hint, only `iters` (= iterations after the block step)
 * and the time do.  Same buffers, checks and stream semantics as wbc_step_batch.  Which kernels run: wbc_plan_tick with warm = 1 -- a call with
 * active_in = NULL runs (and is planned as) the cold tick, which still reports the sets. */
int wbc_step_batch_warm(wbc_solver* s, size_t N, const wbc_batch_in* in, const wbc_batch_out* out, const wbc_observer_state* obs,
                        const int* active_in, int* active_out, void* stream);

/* SURVEY.md 8(f)-1 -- the step Gazebo performs in the reference loop (/root/reference/README.md:58), as the simplest
 * model that closes the loop for rollouts: forward dynamics with the planned GRFs applied,
 *   vdot = M^-1 (S^T tau + Jc^T f + tau_ext - h),  then semi-implicit Euler on (q, v) IN PLACE with the solver's dt.
 * M, h, Jc, tau, f are the outputs of the wbc_step_batch of the same tick (foot lever arms and the own-leg Jacobian
 * blocks are read from Jc).  tau_ext [nv][N] may be NULL. */
int wbc_integrate_batch(wbc_solver* s, size_t N, void* q, void* v, const void* M, const void* h, const void* Jc,
                        const void* tau, const void* f, const void* tau_ext, void* stream);

/* `horizon` dependent ticks of {wbc_step_batch, wbc_integrate_batch} with constant references (BASELINE.json
 * configs[4]: MPC-style WBC-in-the-loop rollouts).  in->q / in->v are ADVANCED IN PLACE (const is cast away);
 * out->tau / out->f must hold the previous tick's outputs (or zeros) on entry and serve as tau_prev / f_prev;
 * out->M, out->h, out->Jc are required.  tau_traj (optional) receives tau of every tick: [horizon][nj][N].
 * On return the out-> buffers (tau, f, status, iters, M, h, Jc, pf), the observer state and (tracking) w_des / vdot_des hold the LAST
 * tick's values, as after per-tick calls; while the call runs their contents are unspecified (rollouts of up to 1024 states are one launch that
 * keeps state, torques and observer inputs on chip from tick to tick and writes them out once). */
int wbc_rollout_batch(wbc_solver* s, size_t N, int horizon, const wbc_batch_in* in, const wbc_batch_out* out,
                      const wbc_observer_state* obs, const void* tau_ext, void* tau_traj, void* stream);

/* SURVEY.md 8(f)-3 -- the caller on the input side of the tick: "a motion planner for the trajectory of the robot's
 * center of mass" (/root/reference/README.md:11) produces the references the whole-body controller tracks.  The
 * planner's source is absent from the reference ([UNVERIFIED] interface); this entry point is the build's definition:
 *   plan [12][N]: c0 (3) start CoM, c1 (3) goal CoM (world), T duration [s], t0 time already elapsed [s],
 *                 quat_des (x,y,z,w) desired trunk attitude
 *   rest-to-rest quintic c_ref(u), u = clamp((t0 + t) / T, 0, 1)   (T <= 0: the goal itself)
 *   a_cmd = cdd_ref + kp_com (c_ref - c) + kd_com (cd_ref - cd)      c, cd = CoM position / velocity of (q, v)
 *   alpha_cmd = kp_rot e_R - kd_rot omega,  e_R = 2 vec(quat_des (x) quat^-1) with non-negative scalar part
 *   vdot_des = [a_cmd; alpha_cmd; kp_joint (q_nom - q_j) - kd_joint qd_j]
 *   w_des = [F; (c - p_b) x F + R diag(inertia_nom) R^T alpha_cmd],  F = m_total (a_cmd - gravity)
 * Writes w_des [6][N] and vdot_des [nv][N] (the wbc_batch_in fields of the same names); com (optional) [6][N] = c, cd. */
typedef struct wbc_ref_params {
  double kp_com[3], kd_com[3];
  double kp_rot[3], kd_rot[3];
  double kp_joint, kd_joint;
  double inertia_nom[3];    /* nominal trunk inertia, body axes */
  double q_nom[WBC_MAXV];   /* nominal joint posture, caller's joint order */
} wbc_ref_params;
#define WBC_PLAN_WORDS 12
void wbc_ref_params_default(wbc_ref_params* g);
int wbc_solver_set_ref_params(wbc_solver* s, const wbc_ref_params* g);
int wbc_reference_batch(wbc_solver* s, size_t N, const void* q, const void* v, const void* plan, double t,
                        void* w_des, void* vdot_des, void* com, void* stream);

/* wbc_rollout_batch with the planner in the loop: every tick k first regenerates in->w_des / in->vdot_des from `plan`
 * at t = k * dt (those two buffers are OVERWRITTEN; const is cast away), then steps and integrates.
 * com_traj (optional) receives the CoM state the planner saw at every tick: [horizon][6][N]. */
int wbc_rollout_tracking_batch(wbc_solver* s, size_t N, int horizon, const wbc_batch_in* in, const wbc_batch_out* out,
                               const wbc_observer_state* obs, const void* tau_ext, const void* plan, void* tau_traj,
                               void* com_traj, void* stream);

/* Single-robot, host-pointer, double-precision convenience call: the shape of the reference's
 * one-robot tick (BASELINE.json configs[0]).  Runs wbc_step_batch with N = 1 on the GPU and
 * synchronises.  obs_integ/obs_r (host, nv each) are in/out and may be NULL when the observer is off. */
int wbc_compute_torques(wbc_solver* s, const double* q, const double* v, const double* w_des,
                        const double* vdot_des, const double* normals, const double* mu, int mask,
                        const double* tau_prev, const double* f_prev, double* obs_integ, double* obs_r,
                        double* tau, double* f, int* status);

/* The single-robot loop without staging copies (fp64 solvers): wbc_one_map hands out HOST pointers into the solver's pinned
 * image -- the caller keeps q, v, references, terrain and the observer state there and rewrites only what changed -- and
 * wbc_one_tick runs one tick on the image in place and returns when tau, f, status, iters (and the updated observer state) are
 * in it.  The wait follows wbc_solver_options.one_zerocopy (2: polled ticket, no hipStreamSynchronize).  Stands for the same
 * per-tick entry point as wbc_compute_torques (name [UNVERIFIED]); one solver = one robot = one image. */
typedef struct wbc_one_image {
  double *q, *v, *w_des, *vdot_des, *normals, *mu, *tau_prev, *f_prev, *obs_integ, *obs_r;   /* inputs (observer state: in/out) */
  double *tau, *f;                                                                           /* outputs */
  int *mask, *status, *iters;
} wbc_one_image;
int wbc_one_map(wbc_solver* s, wbc_one_image* img);
int wbc_one_tick(wbc_solver* s);

/* Observer start-up for the single-robot host-pointer loop (see wbc_observer_state): writes integ = M(q) v and r = 0
 * (host arrays of nv doubles).  Synchronises. */
int wbc_observer_init(wbc_solver* s, const double* q, const double* v, double* obs_integ, double* obs_r);

/* Single-robot, host-pointer form of wbc_reference_batch (q[19], v[18], plan[12] in; w_des[6], vdot_des[18] and the
 * optional com[6] out): the planner call of a one-robot control loop.  Synchronises. */
int wbc_compute_reference(wbc_solver* s, const double* q, const double* v, const double* plan, double t,
                          double* w_des, double* vdot_des, double* com);

/* ---- measurement: per-kernel HIP-event timing on the stream the kernels are launched on (wbc_solver_options.timing_mode:
 * the dispatch's own start / stop events, or an event pair recorded around the launch).  Enabling allocates a ring of
 * 4096 event pairs once; samples beyond it are dropped until wbc_solver_collect_timing drains the ring, so a tick never
 * allocates.  While timing is on, the instrumented dispatches are not hipGraph-capturable. ---- */
int wbc_solver_enable_timing(wbc_solver* s, int on); /* 0 off; 1 every kernel launch; k > 1: every k-th tick */
#define WBC_TIMING_KINDS 6   /* kinds this build knows */
/* synchronises the recorded events; returns summed milliseconds and launch counts since the last reset, indexed
 * 0 = dyn_sweep kernel, 1 = QP kernel, 2 = rnea_step kernel (no-M/h/Jc ticks) / stand-alone observer kernel,
 * 3 = fused tick kernel (sweep + QP of small batches in one launch), 4 = per-lane QP kernel (then 1 = the dense kernel's
 * pass over the states the per-lane kernel handed over), 5 = persistent rollout kernel (one launch = a whole horizon); resets the accumulators. */
int wbc_solver_collect_timing_n(wbc_solver* s, double* ms, int* launches, int cap /* entries of ms / launches: at most cap are written */);
/* the unsized call of ABI <= 6: writes FIVE entries (kinds 0 .. 4), what every version of it has written; use the sized call */
int wbc_solver_collect_timing(wbc_solver* s, double* ms, int* launches);

/* ---- multi-device: ONE host process, one solver per GPU of the node.  The reference is a single C++
 * process (/root/reference/README.md:58-60); this is how that process shards a batch over the node without Python.
 * The batch splits into contiguous slices (wbc_shard_range), shard k runs on devices[k] on its own stream; there is NO
 * data-path collective.  Optional consumer-side collective: every device receives all torques -- RCCL ncclAllGather
 * over xGMI (communicators from ncclCommInitAll; librccl is loaded on demand) or peer copies. ---- */
typedef struct wbc_multi wbc_multi;
enum wbc_gather_backend { WBC_GATHER_NONE = 0, WBC_GATHER_RCCL = 1, WBC_GATHER_PEER_COPY = 2 };
/* contiguous balanced slices: the first (n_total % n_shards) shards hold one extra state */
int wbc_shard_range(size_t n_total, int n_shards, int shard, size_t* start, size_t* count);
/* devices[k] = HIP device of shard k (RCCL: all distinct; NONE / PEER_COPY: a device may host several shards).
 * max_batch_total bounds n_total of every later call; opt may be NULL. */
int wbc_multi_create(const wbc_model* m, const wbc_params* p, int dtype, const int* devices, int n_devices,
                     size_t max_batch_total, int gather_backend, const wbc_solver_options* opt, wbc_multi** out);
void wbc_multi_destroy(wbc_multi* mm);
int wbc_multi_size(const wbc_multi* mm);                    /* number of shards */
int wbc_multi_device(const wbc_multi* mm, int shard);       /* HIP device of a shard */
wbc_solver* wbc_multi_solver(wbc_multi* mm, int shard);     /* the shard's solver (dynamics, timing, reference ... on its device) */
/* hipStream_t the shard's work is enqueued on.  These are hipStreamNonBlocking streams: they do NOT synchronise with the
 * null stream or with any stream of the caller.  Buffers produced elsewhere (copies, fills, a previous consumer) must be
 * complete -- or ordered with hipStreamWaitEvent on this stream -- before wbc_multi_step_batch / wbc_multi_rollout_batch /
 * wbc_multi_allgather_tau are called, and results are visible to other streams only after wbc_multi_synchronize (or an
 * event recorded on this stream). */
void* wbc_multi_stream(wbc_multi* mm, int shard);
int wbc_multi_rccl_ranks(const wbc_multi* mm);              /* ranks of the RCCL communicator; 0 when RCCL is not in use */
int wbc_multi_set_params(wbc_multi* mm, const wbc_params* p);
/* One control tick of n_total states.  in[k] / out[k] / obs[k] (arrays of wbc_multi_size entries; obs may be NULL when
 * the observer is off) describe shard k's slice: device pointers on devices[k], component-major with N = the shard's
 * count.  Enqueues every shard on its stream and returns without synchronising. */
int wbc_multi_step_batch(wbc_multi* mm, size_t n_total, const wbc_batch_in* in, const wbc_batch_out* out,
                         const wbc_observer_state* obs);
/* wbc_step_batch_warm per shard (a sharded closed loop): active[k] = shard k's carried active sets, int32 [count_k] on devices[k],
 * read as the start of this tick's QPs and rewritten in place with the sets they end on (all zero = cold) */
int wbc_multi_step_batch_warm(wbc_multi* mm, size_t n_total, const wbc_batch_in* in, const wbc_batch_out* out,
                              const wbc_observer_state* obs, int* const* active);
/* wbc_rollout_batch per shard (BASELINE.json configs[4]: rank-local for all ticks); tau_ext[k] may be NULL (as may tau_ext) */
int wbc_multi_rollout_batch(wbc_multi* mm, size_t n_total, int horizon, const wbc_batch_in* in, const wbc_batch_out* out,
                            const wbc_observer_state* obs, const void* const* tau_ext);
/* All-gather of the torques behind a tick (on the shard streams): tau_local[k] = shard k's [nj][count_k] on devices[k];
 * tau_all[k] on devices[k] receives wbc_multi_size blocks of nj * count_0 scalars, block j = shard j's tau exactly as
 * shard j laid it out ([nj][count_j], packed; the tail of a shorter shard's block is padding). */
int wbc_multi_allgather_tau(wbc_multi* mm, size_t n_total, const void* const* tau_local, void* const* tau_all);
/* The same gather OFF the tick's critical path: it is enqueued on per-shard gather streams behind the tick that is on the shard streams
 * now and runs BESIDE the next tick.  The caller double-buffers tau (two tau buffers per shard, ticks alternate between them) and names
 * the slot (0 / 1) of the buffer being gathered; wbc_multi_gather_wait(slot) makes every shard stream wait -- on the device, the host
 * returns at once -- for that slot's last gather: call it before the tick that overwrites that slot's tau.  Per tick k, slot b = k & 1:
 *     wbc_multi_gather_wait(mm, b);  wbc_multi_step_batch(... out[b] ...);  wbc_multi_allgather_tau_async(mm, n, tau_b, tau_all_b, b);  */
int wbc_multi_allgather_tau_async(wbc_multi* mm, size_t n_total, const void* const* tau_local, void* const* tau_all, int slot);
int wbc_multi_gather_wait(wbc_multi* mm, int slot);
/* WBC_GATHER_PEER_COPY: where every device can map every other's memory, shard j's block reaches all devices through ONE kernel that stores into the
 * tau_all buffers directly (peer mappings).  That needs every tau_all[k] to be plain hipMalloc memory of devices[k]: the library checks each new set of
 * buffers once (hipPointerGetAttributes: device memory of the right device) and falls back to hipMemcpyPeerAsync per block otherwise -- but memory from a
 * virtual-memory pool (PyTorch expandable segments, hipMallocAsync) can pass that check without being mapped on the peers.  Callers with such buffers
 * switch the push kernel off: on = 1 -> always copies, 0 (default) -> push where allowed.  wbc_multi_gather_pushes: 1 when the next peer gather would push. */
int wbc_multi_set_peer_copies(wbc_multi* mm, int on);
int wbc_multi_gather_pushes(const wbc_multi* mm);
/* One call per tick of that double-buffered loop: wbc_multi_gather_wait(slot), the tick (wbc_multi_step_batch, or _warm when active is given)
 * writing out[k].tau -- the slot's buffer -- and wbc_multi_allgather_tau_async of out[k].tau into tau_all[k]: two tickets to the issue threads. */
int wbc_multi_tick_gather(wbc_multi* mm, size_t n_total, const wbc_batch_in* in, const wbc_batch_out* out, const wbc_observer_state* obs,
                          int* const* active /* may be NULL */, void* const* tau_all, int slot);
int wbc_multi_synchronize(wbc_multi* mm);                   /* waits for every shard stream (and gather stream) */
int wbc_multi_issue_threads(const wbc_multi* mm);           /* number of issue threads (0: the shards are issued on the caller's thread) */
/* host time the caller has spent INSIDE the tick / rollout / gather entry points since the last reset: what feeding the devices costs */
int wbc_multi_host_stats(wbc_multi* mm, unsigned long long* calls, double* seconds, int reset);
/* diagnostics: the time of `iters` EMPTY tickets through the issue threads (what a round trip costs apart from the HIP calls inside it) */
int wbc_multi_probe_issue(wbc_multi* mm, int iters, double* seconds);
/* self-test of the issue threads without a device (run by the CPU test-suite): every thread must run every ticket exactly once, in order, and a thread's
 * error must reach the caller; spin_us = 0 forces the parked (condition variable) path, pause_us > 0 lets the threads park between tickets */
int wbc_multi_selftest_issue(int threads, int tickets, int spin_us, int pause_us);
/* Host-resident batch (a C++ caller that holds host arrays, e.g. the ROS side): in / out / obs hold HOST pointers to
 * component-major arrays [ncomp][n_total] of the solver's scalar type; slices are scattered to the devices with pitched
 * copies, stepped, and tau, f, status, iters (and the observer state, when obs is given) gathered back.  out->M, h, Jc,
 * pf must be NULL.  Synchronises.  PCIe-bound by construction. */
int wbc_multi_step_host(wbc_multi* mm, size_t n_total, const wbc_batch_in* host_in, const wbc_batch_out* host_out,
                        const wbc_observer_state* host_obs);

/* ------------------------------------------------------------------------------------------------------------------------
 * Dense QPs of run-time size -- the general path beside the controller's own 12-variable GRF QP (which wbc_step_batch solves
 * with kernels written around its structure).  Replaces: whatever QP library the reference controller links for its
 * "optimization problem based on the modulation of ground reaction forces" (/root/reference/README.md:11; the source is an
 * absent submodule, .gitmodules:4-6, so its variable set is unknown -- a formulation that also carries accelerations, slacks or
 * joint-torque rows is assembled by the caller and solved here).
 *
 *     min 1/2 x^T H x + g^T x    s.t.   C_i x = d_i (i < meq),   C_i x >= d_i (meq <= i < m);    1 <= n <= 36, 0 <= m <= 64
 *
 * All pointers are DEVICE pointers on the current device, PROBLEM-major (one problem's data contiguous): H [N][n*n] row-major,
 * symmetric positive definite (both triangles given, the lower one is read); g [N][n]; C [N][m*n] row-major; d [N][m]; x [N][n];
 * lambda [N][m] or NULL (multipliers: >= 0 for inequality rows, any sign for equality rows; 0 for inactive rows); status [N]:
 * 0 optimal, 1 iteration limit, 2 infeasible, 3 H not positive definite; iters [N] or NULL.  One QP per wavefront, factors and
 * working set in LDS (Goldfarb-Idnani dual active set); enqueued on hipStream (NULL = the default stream), returns without
 * synchronising.  dtype: WBC_F64 / WBC_F32 = the scalar type of every array and of the arithmetic. */
int wbc_qp_dense_batch(int dtype, size_t N, int n, int m, int meq, const void* H, const void* g, const void* C, const void* d,
                       int max_iter, double tol, void* x, void* lambda, int* status, int* iters, void* hipStream);

const char* wbc_strerror(int status);
const char* wbc_last_error(void); /* thread-local detail string of the last failure */
int wbc_abi_version(void); /* 9 */

#ifdef __cplusplus
}
#endif
#endif /* WBC_HIP_H */
