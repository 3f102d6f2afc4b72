/*
 * uav_oracle.h -- CPU restatement (plain C, double precision) of the
 * reference's PathPlan_City env hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this.  The shipped path
 * (dqn_based_uav_3d_path_planer_amd/csrc/, include/uavenv.h) never does.
 *
 * Parity status: PINNED BY EXECUTION.  The reference has no tests / golden
 * vectors of its own, so the pins are vectors generated
 * by running the reference itself on a scratch copy (oracle/gen_golden.py ->
 * tests/golden/ *.npz).  tests/test_oracle_golden.py checks this file against
 * them bit-for-bit (same glibc libm: every x**2 is pow(x, 2.0) as CPython
 * does, no FMA contraction, same operation order).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference root).
 */
#ifndef UAV_ORACLE_H
#define UAV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_KMAX 128      /* sub-goal capacity of the oracle (reference: unbounded list) */
#define ORC_OBS_DIM 100   /* Agents/UAV.py:517 */

enum { ORC_INFO_NORMAL = 0, ORC_INFO_SUCCESS = 1, ORC_INFO_LOSE = 2 };

/* One cylinder: Obstacles/building.py:6-11 (+ optional velocity read by
 * Agents/UAV.py:180; stock buildings have none -> pass zeros). */
typedef struct {
    double cx, cy, cz, R, H;
    double vx, vy, vz;
} orc_building;

/* World: BaseClass/BaseEnv.py:17-22 + Envs/PathPlan_City.py:41-51 */
typedef struct {
    double len, width, h;
    int32_t nb;
    int32_t _pad;
    orc_building *b;
} orc_world;

/* Python's Mersenne Twister (random.seed(int) / random.random()). */
typedef struct {
    uint32_t mt[624];
    int32_t idx;
    int32_t ext_n, ext_i;      /* when ext != NULL random() replays ext[0..ext_n) instead of the twister */
    const double *ext;
} orc_rng;

/* Agents/UAV.py per-agent parameters + state touched by the hot path. */
typedef struct {
    /* parameters */
    double max_v;            /* UAV.py:25  int(Max_V) */
    double steering_angle;   /* UAV.py:26  radians */
    int32_t max_step;        /* UAV.py:32 */
    int32_t apf_enabled;     /* UAV.py:142 */
    double P_i, v_0, d_0, rho, s, A, P_b, F_b, xi;   /* UAV.py:50-58 (A, xi include the j terms) */
    /* state */
    double px, py, pz;       /* position */
    double vx, vy, vz;       /* V_vector */
    double V;                /* speed magnitude */
    double gx, gy, gz;       /* goal */
    int32_t step;            /* Step */
    int32_t done;            /* self.done (agent done) */
    int32_t n_sub;           /* len(sub_goals) */
    int32_t reach_goal;
    double score, total_score, path_len;
    int64_t train_epoch;
    double v_dir, start2goal, len_astar;
    int32_t error;           /* set when the reference would have raised (KMAX overflow, APF cum_force>100) */
    int32_t sub0_alias;      /* 1 while sub_goals[0] IS the position object: RRT.getPath puts the `start` Loc
                                (== self.position, UAV.py:218 -> RRT.py:69,96-103) at the head of the path, so
                                `position.x += ...` (UAV.py:419-420) moves sub_goals[0] too until the collision
                                branch rebinds position (UAV.py:427) or the sub-goal is popped (UAV.py:469) */
    double sub[ORC_KMAX][3]; /* sub_goals[k] = (x,y,z); sub[0] is the current one */
} orc_uav;

/* ---- geometry / collision ---- */
double orc_calculate_angle(double x1, double y1, double x2, double y2);       /* CalMod.py:89-102 (mod=1) */
double orc_distance(double x1, double y1, double z1, double x2, double y2, double z2); /* CalMod.py:64-65 */
int orc_check_threaten(const orc_building *b, double x, double y, double z);  /* building.py:20-26 */
int orc_threaten_rate(const orc_world *w, double x, double y, double z);      /* PathPlan_City.py:215-223 */
void orc_threaten_rate_many(const orc_world *w, int64_t n, const double *xyz, int32_t *out);

/* ---- agent ---- */
double orc_calc_v(orc_uav *u);                                                /* UAV.py:246-253 */
double orc_calc_fly_power(orc_uav *u);                                        /* UAV.py:239-245 */
int orc_cal_force(const orc_world *w, orc_uav *u, double x, double y, double z, double f[3]); /* UAV.py:174-210 */
void orc_update_pathplan(const orc_world *w, orc_uav *u, double a0,
                         double *reward, int32_t *ret_done, int32_t *info);   /* UAV.py:397-513 */
void orc_state_pathplan(const orc_world *w, const orc_uav *u, double *obs);   /* UAV.py:515-567 */

/* ---- reset + RRT ---- */
void orc_rng_seed(orc_rng *r, uint64_t seed);          /* random.seed(int) */
void orc_rng_external(orc_rng *r, const double *u, int32_t n);  /* replay a given U[0,1) stream (GPU-planner parity) */
double orc_rng_random(orc_rng *r);                     /* random.random() */
double orc_rng_uniform(orc_rng *r, double a, double b);/* random.uniform(a,b) */
/* PathPlan/RRT.py:63-105; returns the number of path nodes written (<= cap), or -needed if cap too small */
int orc_rrt_get_path(const orc_world *w, orc_rng *r, double step_size, int max_iter, double obstacle_step,
                     const double start[3], const double goal[3], double *path_xyz, int cap, int *n_iters);
void orc_reset(const orc_world *w, orc_uav *u, orc_rng *r, double sub_granularity); /* UAV.py:327-366 */

/* ---- batched helpers (CPU baseline; OpenMP over independent agents) ---- */
void orc_step_many(const orc_world *w, orc_uav *u, int64_t n, const double *a0,
                   double *reward, int32_t *ret_done, int32_t *info, double *obs /* n*100 or NULL */,
                   int nthreads);
/* n_steps of update + state per agent inside C, auto-reset from a scenario bank (see uav_oracle.c); returns agent-steps */
int64_t orc_rollout_many(const orc_world *w, orc_uav *u, int64_t n, int32_t n_steps, const double *bank_start_goal,
                         const double *bank_sub, const int32_t *bank_nsub, int32_t bank_m, int32_t bank_k,
                         uint64_t seed, double *obs, double *reward_sum, int nthreads);
int orc_sizeof_uav(void);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
