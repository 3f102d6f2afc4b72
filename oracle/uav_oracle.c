/*
 * uav_oracle.c -- see uav_oracle.h.  TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Arithmetic rules that make this bit-identical to the executed reference on
 * the same glibc:
 *   - CPython evaluates x**2 on floats as libm pow(x, 2.0) (Objects/floatobject.c
 *     float_pow), which is NOT always == x*x (0.08 % of inputs differ by 1 ulp),
 *     so SQ(x) is pow(x, 2.0).
 *   - math.degrees(x) is x * (180.0 / pi); (a + 360) % 360 on positive
 *     operands is fmod.
 *   - compile with -ffp-contract=off and no -ffast-math (see oracle/Makefile).
 *   - the one place the reference goes through numpy/OpenBLAS (RRT.steer's
 *     np.linalg.norm -> ddot, PathPlan/RRT.py:39-46) rounds as a sequential
 *     FMA chain on this platform; norm3_blas() restates that.
 */
#include "uav_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORC_FAST_SQ
/* speed build for the timed CPU baseline: x*x instead of pow(x,2.0) (<=1 ulp apart) */
#define SQ(x) ((x) * (x))
#else
#define SQ(x) pow((x), 2.0)
#endif

static const double ORC_PI = 3.141592653589793;          /* math.pi */
static const double ORC_RAD2DEG = 180.0 / 3.141592653589793; /* CPython mathmodule.c radToDeg */

/* ------------------------------------------------------------------ geometry */

/* BaseClass/CalMod.py:64-65  Eu_Loc_distance(loc1, loc2) */
double orc_distance(double x1, double y1, double z1, double x2, double y2, double z2)
{
    return sqrt(SQ(x1 - x2) + SQ(y1 - y2) + SQ(z1 - z2));
}

/* BaseClass/CalMod.py:89-102  calculate_angle(p1, p2, mod=1): heading p1->p2 in [0, 2pi] */
double orc_calculate_angle(double x1, double y1, double x2, double y2)
{
    double dx = x2 - x1;
    double dy = y2 - y1;
    double angle = atan2(dy, dx);
    angle = angle * ORC_RAD2DEG;                 /* math.degrees */
    double m = fmod(angle + 360.0, 360.0);       /* (angle + 360) % 360, operands >= 0 */
    return m / 180.0 * ORC_PI;
}

/* Obstacles/building.py:20-26 */
int orc_check_threaten(const orc_building *b, double x, double y, double z)
{
    if (z > b->H) return 0;
    /* Eu_Loc_distance(Loc(x, y, self.position.z), self.position) < _R */
    if (orc_distance(x, y, b->cz, b->cx, b->cy, b->cz) < b->R) return 1;
    return 0;
}

/* Envs/PathPlan_City.py:215-223 -- note: `width` bounds BOTH x and y; edges inclusive */
int orc_threaten_rate(const orc_world *w, double x, double y, double z)
{
    if (x < 0 || x > w->width || y < 0 || y > w->width || z < 0 || z > w->h) return 1;
    for (int i = 0; i < w->nb; ++i)
        if (orc_check_threaten(&w->b[i], x, y, z) > 0) return 1;
    return 0;
}

void orc_threaten_rate_many(const orc_world *w, int64_t n, const double *xyz, int32_t *out)
{
    for (int64_t i = 0; i < n; ++i)
        out[i] = orc_threaten_rate(w, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}

/* ------------------------------------------------------------------ agent */

/* Agents/UAV.py:246-253 */
double orc_calc_v(orc_uav *u)
{
    double V = orc_distance(0, 0, 0, u->vx, u->vy, u->vz);
    if (V > u->max_v) {
        u->vx = u->vx * (u->max_v / V);
        u->vy = u->vy * (u->max_v / V);
        V = u->max_v;
    }
    return V;
}

/* Agents/UAV.py:239-245 */
double orc_calc_fly_power(orc_uav *u)
{
    u->V = orc_calc_v(u);
    double V = u->V;
    double induced = u->P_i * sqrt(sqrt(1 + pow(V, 4.0) / (4 * pow(u->v_0, 4.0))) - SQ(V) / (2 * SQ(u->v_0)));
    double parasite = 0.5 * u->d_0 * u->rho * u->s * u->A * pow(V, 3.0);
    double blade = u->xi * u->P_b * (1 + 3 * SQ(V) / SQ(u->F_b));
    return induced + parasite + blade;
}

/* Agents/UAV.py:174-210.  Returns 0 and the summed force, or 1 if the
 * reference would have called Cal_SubTask_Dynamic() (which raises). */
int orc_cal_force(const orc_world *w, orc_uav *u, double x, double y, double z, double f[3])
{
    double cum = 0;
    double tx = 0, ty = 0, tz = 0;
    (void)u;
    for (int i = 0; i < w->nb; ++i) {
        const orc_building *b = &w->b[i];
        if (b->vx == 0 && b->vy == 0 && b->vz == 0) continue;          /* :180-182 */
        double dis = orc_distance(x, y, z, b->cx, b->cy, b->cz);        /* :183 */
        double dis2edge = dis - b->R;                                    /* :185 */
        if (dis2edge > 60) continue;                                     /* :186 */
        double v = orc_distance(0, 0, 0, b->vx, b->vy, b->vz);           /* :189 */
        double q = 1 * b->R / (dis2edge * dis2edge);
        double f1 = (q < 1) ? q : 1;                                     /* min(1, ...) :190 */
        double f1_seta = orc_calculate_angle(x, y, b->cx, b->cy);        /* :191 */
        double v_seta = orc_calculate_angle(0, 0, b->vx, b->vy);         /* :193 */
        if (dis2edge < 0) f1 = (-dis2edge > 2) ? -dis2edge : 2;          /* max(-d, 2) :196-197 */
        double f1x = -f1 * cos(f1_seta), f1y = -f1 * sin(f1_seta);       /* :198 */
        double q2 = v * b->R / (dis2edge * dis2edge);
        double f2 = (q2 < 1) ? q2 : 1;                                   /* :200 */
        double f2x = f2 * cos(v_seta), f2y = f2 * sin(v_seta);           /* :201 */
        cum += (f1 + f2);                                                /* :202 */
        tx = (tx + f1x) + f2x;                                           /* :203 */
        ty = (ty + f1y) + f2y;
        tz = (tz + 0) + 0;
        if (cum > 100) { f[0] = tx; f[1] = ty; f[2] = tz; return 1; }    /* :205-208 raises in the reference */
    }
    f[0] = tx; f[1] = ty; f[2] = tz;
    return 0;
}

static void pop_subgoal(orc_uav *u)
{
    u->sub0_alias = 0;
    memmove(&u->sub[0][0], &u->sub[1][0], sizeof(double) * 3 * (size_t)(u->n_sub - 1));
    u->n_sub -= 1;
}

/* Agents/UAV.py:397-513  update_PathPlan(action) with a0 = float(action[0]) */
void orc_update_pathplan(const orc_world *w, orc_uav *u, double a0,
                         double *reward, int32_t *ret_done, int32_t *info)
{
    double r = 0;
    if (u->n_sub == 0) {                                                 /* :400-406 */
        u->done = 1;
        r += (double)(u->max_step - u->step);
        u->score += r;
        *reward = r; *ret_done = 1; *info = ORC_INFO_SUCCESS;
        return;
    }
    u->step += 1;                                                        /* :408 */
    double ox = u->px, oy = u->py, oz = u->pz;                           /* :409 */
    double seta_old = orc_calculate_angle(0, 0, u->vx, u->vy);           /* :411 */
    double dis_old = orc_distance(u->px, u->py, u->pz, u->sub[0][0], u->sub[0][1], u->sub[0][2]);
    double dis2goal_old = orc_distance(u->px, u->py, u->pz, u->gx, u->gy, u->gz);
    double seta_new = seta_old + a0 * u->steering_angle;                 /* :414 */
    u->vx = u->max_v * cos(seta_new);                                    /* :415 */
    u->vy = u->max_v * sin(seta_new);                                    /* :416 */
    u->V = orc_calc_v(u);                                                /* :417 */
    u->px += u->vx;                                                      /* :419 */
    u->py += u->vy;                                                      /* :420 */
    if (u->sub0_alias) {               /* sub_goals[0] is the same Loc object as position: it moved too */
        u->sub[0][0] = u->px; u->sub[0][1] = u->py; u->sub[0][2] = u->pz;
    }
    double tri_goal = orc_calculate_angle(u->px, u->py, u->sub[0][0], u->sub[0][1]);   /* :422 */
    double tri_V = orc_calculate_angle(0, 0, u->vx, u->vy);              /* :423 */
    if (orc_threaten_rate(w, u->px, u->py, u->pz) == 1) {                /* :425-428 */
        r -= 0.3;
        u->px = ox; u->py = oy; u->pz = oz;   /* rebinds position to the copy: the alias ends here */
        u->sub0_alias = 0;
        tri_V = orc_calculate_angle(u->px, u->py, u->sub[0][0], u->sub[0][1]);
    }
    double dis_new = orc_distance(u->px, u->py, u->pz, u->sub[0][0], u->sub[0][1], u->sub[0][2]);
    double dis2goal_new = orc_distance(u->px, u->py, u->pz, u->gx, u->gy, u->gz);
    r -= 0.13 * fabs(a0);                                                /* :434 */
    r += 0.2 * cos(fabs(tri_goal - tri_V));                              /* :435 */
    r += 0.4 * (dis_old - dis_new);                                      /* :436 */
    r += 0.4 * (dis2goal_old - dis2goal_new);                            /* :437 */
    r -= 0.1;                                                            /* :438 */
    r -= 0.01 * fabs(u->pz - u->sub[0][2]);                              /* :439-440 */
    u->path_len += u->V;                                                 /* :443 */
    u->train_epoch += 1;                                                 /* :444 */

    if (u->apf_enabled == 1) {                                           /* :448-453 */
        for (int k = 0; k < u->n_sub; ++k) {                             /* Adjust_subgoal :156-166 */
            double f[3];
            if (orc_cal_force(w, u, u->sub[k][0], u->sub[k][1], u->sub[k][2], f)) u->error = 2;
            u->sub[k][0] += f[0]; u->sub[k][1] += f[1]; u->sub[k][2] += f[2];
        }
        u->sub0_alias = 0;   /* Adjust_subgoal rebuilds the list from NEW Loc objects (:162-165): the alias ends */
        double f[3];
        if (orc_cal_force(w, u, u->px, u->py, u->pz, f)) u->error = 2;
        double force = orc_distance(0, 0, 0, f[0], f[1], f[2]);
        double tri_force = orc_calculate_angle(0, 0, f[0], f[1]);
        r += 0.2 * force * cos(fabs(tri_force - tri_V));
    }

    double d_sub = orc_distance(u->px, u->py, u->pz, u->sub[0][0], u->sub[0][1], u->sub[0][2]);
    if (u->step >= u->max_step) {                                        /* :456-465 */
        u->done = 1;
        r += (50 - d_sub);
        u->score += r;
        u->total_score += r;
        *reward = r; *ret_done = 1; *info = ORC_INFO_LOSE;
        return;
    }
    if (d_sub < 7 ||
        (orc_distance(u->px, u->py, u->pz, u->gx, u->gy, u->gz) <
         orc_distance(u->sub[0][0], u->sub[0][1], u->sub[0][2], u->gx, u->gy, u->gz))) {   /* :466 */
        r += (50 - d_sub);                                               /* :468 */
        pop_subgoal(u);                                                  /* :469 */
        if (u->n_sub == 0) {                                             /* :470-483 */
            r += 50;
            u->done = 1;
            r += (double)(u->max_step - u->step);
            u->score += r;
            u->reach_goal = 1;
            u->total_score += r;
            *reward = r; *ret_done = 1; *info = ORC_INFO_SUCCESS;
            return;
        }
        /* reset("local reset") :486, :328-332 */
        u->step = 0;
        u->score = 0;
        u->V = orc_calc_v(u);
        tri_goal = orc_calculate_angle(u->px, u->py, u->sub[0][0], u->sub[0][1]);   /* :488 */
        tri_V = orc_calculate_angle(0, 0, u->vx, u->vy);                 /* :489 */
        r += 0.2 * cos(fabs(tri_goal - tri_V));                          /* :490 */
        r += (double)(u->max_step - u->step);                            /* :491 */
        u->score += r;
        u->total_score += r;
        *reward = r; *ret_done = 1; *info = ORC_INFO_SUCCESS;            /* returned done, agent NOT done */
        return;
    }
    if (orc_distance(u->px, u->py, u->pz, u->gx, u->gy, u->gz) < 7) {    /* :496-509 */
        u->done = 1;
        r += 50;
        r += (double)(u->max_step - u->step);
        u->score += r;
        u->reach_goal = 1;
        u->total_score += r;
        *reward = r; *ret_done = 1; *info = ORC_INFO_SUCCESS;
        return;
    }
    u->score += r;                                                       /* :511-513 */
    u->total_score += r;
    *reward = r; *ret_done = 0; *info = ORC_INFO_NORMAL;
}

/* Agents/UAV.py:515-567  state_PathPlan() -> float64[100] */
void orc_state_pathplan(const orc_world *w, const orc_uav *u, double *o)
{
    for (int i = 0; i < ORC_OBS_DIM; ++i) o[i] = 0.0;
    o[0] = (double)u->step / 100;                                        /* :518 */
    if (u->n_sub >= 1) {                                                 /* :519-522 */
        o[1] = (u->sub[0][0] - u->px) / 10;
        o[2] = (u->sub[0][1] - u->py) / 10;
        o[3] = (u->sub[0][2] - u->pz) / 10;
    }
    o[4] = u->V;                                                         /* :523 */
    o[5] = u->vx;
    o[6] = u->vy;
    o[7] = orc_calculate_angle(0, 0, u->vx, u->vy);                      /* :526 */
    if (u->n_sub >= 2) {                                                 /* :528-531 */
        o[8] = (u->sub[1][0] - u->px) / 10;
        o[9] = (u->sub[1][1] - u->py) / 10;
        o[10] = (u->sub[1][2] - u->pz) / 10;
    }
    static const double scale[3] = {1.0, 5.0, 10.0};                     /* :533-555 */
    for (int s = 0; s < 3; ++s)
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                double dx = (double)(i - 2), dy = (double)(j - 2);
                double tx = (s == 0) ? u->px + dx : u->px + scale[s] * dx;
                double ty = (s == 0) ? u->py + dy : u->py + scale[s] * dy;
                o[11 + 25 * s + 5 * i + j] = (double)orc_threaten_rate(w, tx, ty, u->pz);
            }
    o[86] = (u->gx - u->px) / 10;                                        /* :557-559 */
    o[87] = (u->gy - u->py) / 10;
    o[88] = (u->gz - u->pz) / 10;
    o[89] = u->pz / 10;                                                  /* :560 */
    for (int k = 1; k <= 5; ++k)                                         /* :562-566 */
        o[89 + k] = (double)orc_threaten_rate(w, u->px, u->py, u->pz - (double)k);
}

/* ------------------------------------------------------------------ Mersenne Twister (CPython _randommodule.c) */

static void mt_init_genrand(orc_rng *r, uint32_t s)
{
    r->mt[0] = s;
    for (int i = 1; i < 624; ++i)
        r->mt[i] = 1812433253U * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->idx = 624;
}

static void mt_init_by_array(orc_rng *r, const uint32_t *key, int len)
{
    mt_init_genrand(r, 19650218U);
    int i = 1, j = 0;
    int k = (624 > len) ? 624 : len;
    for (; k; --k) {
        r->mt[i] = (r->mt[i] ^ ((r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
        ++i; ++j;
        if (i >= 624) { r->mt[0] = r->mt[623]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = 623; k; --k) {
        r->mt[i] = (r->mt[i] ^ ((r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
        ++i;
        if (i >= 624) { r->mt[0] = r->mt[623]; i = 1; }
    }
    r->mt[0] = 0x80000000U;
}

void orc_rng_seed(orc_rng *r, uint64_t seed)
{
    uint32_t key[2];
    int len = 1;
    key[0] = (uint32_t)(seed & 0xffffffffU);
    key[1] = (uint32_t)(seed >> 32);
    if (key[1]) len = 2;
    mt_init_by_array(r, key, len);
    r->ext = 0; r->ext_n = 0; r->ext_i = 0;
}

static uint32_t mt_genrand(orc_rng *r)
{
    static const uint32_t mag01[2] = {0x0U, 0x9908b0dfU};
    uint32_t y;
    if (r->idx >= 624) {
        int kk;
        for (kk = 0; kk < 624 - 397; ++kk) {
            y = (r->mt[kk] & 0x80000000U) | (r->mt[kk + 1] & 0x7fffffffU);
            r->mt[kk] = r->mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1U];
        }
        for (; kk < 623; ++kk) {
            y = (r->mt[kk] & 0x80000000U) | (r->mt[kk + 1] & 0x7fffffffU);
            r->mt[kk] = r->mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1U];
        }
        y = (r->mt[623] & 0x80000000U) | (r->mt[0] & 0x7fffffffU);
        r->mt[623] = r->mt[396] ^ (y >> 1) ^ mag01[y & 1U];
        r->idx = 0;
    }
    y = r->mt[r->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}

void orc_rng_external(orc_rng *r, const double *u, int32_t n)
{
    r->ext = u;
    r->ext_n = n;
    r->ext_i = 0;
}

double orc_rng_random(orc_rng *r)
{
    if (r->ext) return r->ext_i < r->ext_n ? r->ext[r->ext_i++] : 0.75;   /* exhausted stream: keep sampling space */
    uint32_t a = mt_genrand(r) >> 5, b = mt_genrand(r) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

double orc_rng_uniform(orc_rng *r, double a, double b)
{
    return a + (b - a) * orc_rng_random(r);   /* Lib/random.py uniform() */
}

/* ------------------------------------------------------------------ RRT (PathPlan/RRT.py) */

typedef struct { double x, y, z; int parent; double cost; } rrt_node;

/* Loc.distance -- CalMod.py:41-42 */
static double loc_distance(double x1, double y1, double z1, double x2, double y2, double z2)
{
    return sqrt(SQ(x1 - x2) + SQ(y1 - y2) + SQ(z1 - z2));
}

/* np.linalg.norm of a 3-vector == sqrt(ddot(x,x)); OpenBLAS's ddot tail on
 * this platform rounds as fma(x2,x2, fma(x1,x1, x0*x0)) (verified against
 * numpy 2.2 / OpenBLAS 0.3.29 here: 50 000 / 50 000 identical). */
static double norm3_blas(double x0, double x1, double x2)
{
    return sqrt(fma(x2, x2, fma(x1, x1, x0 * x0)));
}

/* RRT.py:48-56 */
static int rrt_obstacle_free(const orc_world *w, const double a[3], const double b[3], double step_size)
{
    int steps = (int)(loc_distance(a[0], a[1], a[2], b[0], b[1], b[2]) / step_size);
    for (int i = 0; i < steps + 1; ++i) {
        double x = a[0] + (b[0] - a[0]) * i / (steps + 1);
        double y = a[1] + (b[1] - a[1]) * i / (steps + 1);
        double z = a[2] + (b[2] - a[2]) * i / (steps + 1);
        if (orc_threaten_rate(w, x, y, z) == 1) return 0;
    }
    return 1;
}

/* RRT.py:63-105 */
int orc_rrt_get_path(const orc_world *w, orc_rng *r, double step_size, int max_iter, double obstacle_step,
                     const double start[3], const double goal[3], double *path_xyz, int cap, int *n_iters)
{
    rrt_node *nodes = (rrt_node *)malloc(sizeof(rrt_node) * (size_t)(max_iter + 2));
    int nn = 0;
    nodes[nn].x = start[0]; nodes[nn].y = start[1]; nodes[nn].z = start[2];
    nodes[nn].parent = -1; nodes[nn].cost = 0.0; nn++;
    int goal_parent = -1;
    double goal_radius = step_size;
    int it;
    for (it = 0; it < max_iter; ++it) {
        /* get_random_point :26-34 */
        double rp[3];
        if (orc_rng_uniform(r, 0, 1) > 0.5) {
            rp[0] = orc_rng_uniform(r, 0, w->len);
            rp[1] = orc_rng_uniform(r, 0, w->width);
            rp[2] = orc_rng_uniform(r, 0, w->h);
        } else {
            rp[0] = goal[0]; rp[1] = goal[1]; rp[2] = goal[2];
        }
        /* nearest_node :36-37 (first minimum) */
        int near = 0;
        double best = loc_distance(nodes[0].x, nodes[0].y, nodes[0].z, rp[0], rp[1], rp[2]);
        for (int i = 1; i < nn; ++i) {
            double d = loc_distance(nodes[i].x, nodes[i].y, nodes[i].z, rp[0], rp[1], rp[2]);
            if (d < best) { best = d; near = i; }
        }
        /* steer :39-46 */
        double dir[3] = {rp[0] - nodes[near].x, rp[1] - nodes[near].y, rp[2] - nodes[near].z};
        double length = norm3_blas(dir[0], dir[1], dir[2]);
        double nl[3];
        if (length < step_size) {
            nl[0] = rp[0]; nl[1] = rp[1]; nl[2] = rp[2];
        } else {
            dir[0] = dir[0] / length; dir[1] = dir[1] / length; dir[2] = dir[2] / length;
            nl[0] = nodes[near].x + dir[0] * step_size;
            nl[1] = nodes[near].y + dir[1] * step_size;
            nl[2] = nodes[near].z + dir[2] * step_size;
        }
        double na[3] = {nodes[near].x, nodes[near].y, nodes[near].z};
        if (!rrt_obstacle_free(w, na, nl, obstacle_step)) continue;        /* :77-78 */
        int me = nn;
        nodes[me].x = nl[0]; nodes[me].y = nl[1]; nodes[me].z = nl[2];
        nodes[me].parent = near;
        nodes[me].cost = nodes[near].cost + loc_distance(na[0], na[1], na[2], nl[0], nl[1], nl[2]);
        nn++;
        for (int i = 0; i < nn; ++i) {                                      /* :84-90 */
            if (i == me) continue;
            double d = loc_distance(nodes[i].x, nodes[i].y, nodes[i].z, nl[0], nl[1], nl[2]);
            if (d < step_size && nodes[me].cost > nodes[i].cost + d) {
                double pa[3] = {nodes[i].x, nodes[i].y, nodes[i].z};
                if (rrt_obstacle_free(w, pa, nl, obstacle_step)) {
                    nodes[me].parent = i;
                    nodes[me].cost = nodes[i].cost + loc_distance(pa[0], pa[1], pa[2], nl[0], nl[1], nl[2]);
                }
            }
        }
        if (loc_distance(nl[0], nl[1], nl[2], goal[0], goal[1], goal[2]) <= goal_radius) {   /* :92-94 */
            goal_parent = me;
            ++it;
            break;
        }
    }
    if (n_iters) *n_iters = it;
    /* path :96-103 : goal, then parents back to the start, reversed */
    int count = 1;
    for (int p = goal_parent; p >= 0; p = nodes[p].parent) count++;
    if (count > cap) { free(nodes); return -count; }
    int k = count - 1;
    path_xyz[3 * k] = goal[0]; path_xyz[3 * k + 1] = goal[1]; path_xyz[3 * k + 2] = goal[2];
    for (int p = goal_parent; p >= 0; p = nodes[p].parent) {
        --k;
        path_xyz[3 * k] = nodes[p].x; path_xyz[3 * k + 1] = nodes[p].y; path_xyz[3 * k + 2] = nodes[p].z;
    }
    free(nodes);
    return count;
}

/* Agents/UAV.py:327-366 (global reset) */
void orc_reset(const orc_world *w, orc_uav *u, orc_rng *r, double sub_granularity)
{
    u->step = 0;
    u->score = 0;
    u->done = 0;
    double seta = orc_rng_uniform(r, 0, 2 * ORC_PI);                    /* :344 */
    u->v_dir = seta;
    u->vx = u->max_v * cos(seta);
    u->vy = u->max_v * sin(seta);
    u->vz = 0;
    u->V = orc_calc_v(u);
    double x = orc_rng_uniform(r, 10, 210);                              /* :353-355 */
    double y = orc_rng_uniform(r, 1, 10);
    u->px = x; u->py = y; u->pz = 0;
    x = orc_rng_uniform(r, 330, 490);                                    /* :356-358 */
    y = orc_rng_uniform(r, 420, 490);
    u->gx = x; u->gy = y; u->gz = 0;
    double s[3] = {u->px, u->py, u->pz}, g[3] = {u->gx, u->gy, u->gz};
    /* Cal_SubTask :216-225 -> RRTPlanner(step=sub_granularity, iter 10000, obstacle_step 5) */
    int n = orc_rrt_get_path(w, r, sub_granularity, 10000, 5.0, s, g, &u->sub[0][0], ORC_KMAX, 0);
    if (n < 0) { u->error = 1; n = 0; }
    u->n_sub = n;
    u->sub0_alias = (n >= 2) ? 1 : 0;   /* path[0] is the start node == the position object (RRT.py:69) */
    u->total_score = 0;
    u->path_len = 0;
    u->reach_goal = 0;
    u->start2goal = orc_distance(u->px, u->py, u->pz, u->gx, u->gy, u->gz);
    double L = 0;                                                         /* calculate_path_len CalMod.py:133-139 */
    for (int k = 1; k < n; ++k)
        L += orc_distance(u->sub[k - 1][0], u->sub[k - 1][1], u->sub[k - 1][2],
                          u->sub[k][0], u->sub[k][1], u->sub[k][2]);
    u->len_astar = L;
}

/* ------------------------------------------------------------------ batch (CPU baseline) */

void orc_step_many(const orc_world *w, orc_uav *u, int64_t n, const double *a0,
                   double *reward, int32_t *ret_done, int32_t *info, double *obs, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) {
        orc_update_pathplan(w, &u[i], a0[i], &reward[i], &ret_done[i], &info[i]);
        if (obs) orc_state_pathplan(w, &u[i], obs + (size_t)ORC_OBS_DIM * (size_t)i);
    }
}

/* The timed CPU baseline (bench.py cpu_baseline): n_steps rounds of update_PathPlan + state_PathPlan for n
 * independent agents, entirely inside C -- each OpenMP thread owns a contiguous block of agents and runs all its
 * steps without returning to the caller (no fork/join, no interpreter per step).  Steering actions come from a
 * per-agent 64-bit LCG (uniform in [-1, 1): what the random policy of the GPU run does); an agent whose episode ended
 * (Check_uav_Done, PathPlan_City.py:252-259) restarts from scenario `bank` row (agent + episode count) mod m with the
 * next LCG draw as heading -- the CPU twin of the GPU loop's auto-reset from the same scenario bank (UAV.reset with
 * the RRT result pre-planned, UAV.py:327-366).  obs: n * 100 doubles (each agent's row is overwritten every step).
 * Returns the number of agent-steps executed. */
int64_t orc_rollout_many(const orc_world *w, orc_uav *u, int64_t n, int32_t n_steps, const double *bank_start_goal,
                         const double *bank_sub, const int32_t *bank_nsub, int32_t bank_m, int32_t bank_k,
                         uint64_t seed, double *obs, double *reward_sum, int nthreads)
{
    int64_t total = 0;
    double rsum = 0.0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) reduction(+ : total, rsum)
#endif
    for (int64_t i = 0; i < n; ++i) {
        orc_uav *a = &u[i];
        uint64_t x = seed * 0x9E3779B97F4A7C15ull + (uint64_t)i * 0xD1342543DE82EF95ull + 1ull;
        int64_t episodes = 0;
        for (int32_t t = 0; t < n_steps; ++t) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const double a0 = (double)(x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
            double r;
            int32_t d, info;
            orc_update_pathplan(w, a, a0, &r, &d, &info);
            rsum += r;
            if (a->done && bank_m > 0) {
                ++episodes;
                const int64_t sc = (i + episodes * 7919) % bank_m;
                const double *sg = bank_start_goal + 6 * sc;
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                const double seta = (double)(x >> 11) * (2.0 * ORC_PI / 9007199254740992.0);
                a->step = 0; a->score = 0; a->done = 0;
                a->v_dir = seta;
                a->vx = a->max_v * cos(seta); a->vy = a->max_v * sin(seta); a->vz = 0;
                a->V = orc_calc_v(a);
                a->px = sg[0]; a->py = sg[1]; a->pz = sg[2];
                a->gx = sg[3]; a->gy = sg[4]; a->gz = sg[5];
                int32_t ns = bank_nsub[sc];
                if (ns > ORC_KMAX) ns = ORC_KMAX;
                memcpy(&a->sub[0][0], bank_sub + (size_t)sc * (size_t)bank_k * 3, (size_t)ns * 3 * sizeof(double));
                a->n_sub = ns;
                a->sub0_alias = ns >= 2 ? 1 : 0;
                a->total_score = 0; a->path_len = 0; a->reach_goal = 0;
            }
            if (obs) orc_state_pathplan(w, a, obs + (size_t)ORC_OBS_DIM * (size_t)i);
        }
        total += n_steps;
    }
    if (reward_sum) *reward_sum = rsum;
    return total;
}

int orc_sizeof_uav(void) { return (int)sizeof(orc_uav); }

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
