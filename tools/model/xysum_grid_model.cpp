// xysum_grid_model.cpp -- CPU model of the wave-parallel, reference-ORDER accumulation of LinearFit::xySum
// (reference cpp/psk_soft.cpp:70-79) used by fit_block in psk_fast_loop.h (xysum_grid), checked against the
// plain sequential recurrence on random blocks.  Development aid: it shows which blocks the method covers
// and that what it produces there is bit-identical; on the GPU nothing rests on that -- every candidate is
// verified against the defining recurrence (fit_sums_verify) and the lane-after-lane chain takes the rest.
//
// Reference, per symbol j of a block (steady state: the window is full):
//     r_j = fl64(s_{j-1} - c_j)      c_j = fl64(xdelta * ySum)  (53 significant bits)
//     s_j = fl64(r_j + t_j)          t_j = (double) float term   (24 significant bits)
// Let [2^p, 2^(p+1)) be the binade of the intermediates r_j and q = 2^(p-52) its ulp; everything below is
// in units of q (scaling by a power of two is exact).
//   mode A  |s| < 2^(p+1): s_{j-1} is a multiple of q, so r_j = s_{j-1} - RN(c_j) whatever the state,
//           except on an exact tie, which goes to the even neighbour: that depends on the parity of s_{j-1}.
//           r_j + t_j is exact.  After a tie the parity is known (even + t_j): a segmented xor scan gives the
//           parity at every position, the tie corrections follow, ONE exact prefix sum yields all s_j.
//           (Also covers sums near zero, |s| << |c|: there c is on the grid itself and nothing rounds.)
//   mode B  2^(p+1) <= |s| < 2^(p+2) (the sums sit just above a power of two, the intermediates dip below
//           it): s_{j-1} is an even multiple of q, so the first rounding is settled locally; the second one,
//           to multiples of 2, is exact for an even r_j + t_j and a state-dependent tie for an odd one.
//
//   g++ -O2 -std=c++17 -ffp-contract=off tools/model/xysum_grid_model.cpp -o /tmp/xysum_model && /tmp/xysum_model
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

static const int B = 128;

static inline uint64_t bits(double x)
{
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
static inline int odd(double n) { return std::fabs(n * 0.5 - std::floor(n * 0.5)) != 0.0; }  // n integer-valued

struct Result {
    int mode;  // 0 A, 1 B
    double s[B];
};

static void seq(double s0, const double *c, const double *t, int nvalid, double *s)
{
    double x = s0;
    for (int j = 0; j < nvalid; j++) {
        double r = x - c[j];
        x = r + t[j];
        s[j] = x;
    }
}

// the candidates (every step below is elementwise or a scan on the device)
static Result grid(double s0, const double *c_in, const double *t_in, int nvalid)
{
    Result R;
    const double r_first = s0 - c_in[0];
    int e;
    (void)frexp(r_first, &e);
    const int p = e - 1;
    const double inv_q = ldexp(1.0, 52 - p), q = ldexp(1.0, p - 52);
    const bool modeB = std::fabs(s0) >= ldexp(1.0, p + 1);
    R.mode = modeB;
    bool T[B];
    int V[B], E[B];
    double inc[B], d0[B], d1[B];
    for (int j = 0; j < B; j++) {
        const double c = j < nvalid ? c_in[j] : 0.0, t = j < nvalid ? t_in[j] : 0.0;
        const double nc = c * inv_q, nt = t * inv_q;
        const double ch = std::nearbyint(nc), d = nc - ch;
        const bool tie_r = std::fabs(d) == 0.5;
        const int c_odd = odd(ch);
        if (!modeB) {
            inc[j] = nt - ch;
            T[j] = tie_r;
            E[j] = c_odd;
            V[j] = tie_r ? (odd(inc[j]) ^ c_odd) : odd(inc[j]);
            d0[j] = 0.0;
            d1[j] = d > 0 ? -1.0 : 1.0;
        } else {
            const double kap = (tie_r && c_odd) ? (d > 0 ? -1.0 : 1.0) : 0.0;
            const double u = (nt - ch) + kap;
            T[j] = odd(u);
            E[j] = T[j] ? odd((u - 1.0) * 0.5) : 0;
            V[j] = T[j] ? 0 : odd(u * 0.5);
            inc[j] = u;
            d0[j] = -1.0;
            d1[j] = 1.0;
        }
    }
    const double S0 = s0 * inv_q;
    int P = modeB ? odd(S0 * 0.5) : odd(S0);
    double run = S0;
    for (int j = 0; j < B; j++) {
        double add = inc[j];
        if (T[j]) {
            add += (P ^ E[j]) ? d1[j] : d0[j];
            P = V[j];
        } else {
            P ^= V[j];
        }
        run += add;
        R.s[j] = run * q;
    }
    return R;
}

int main(int argc, char **argv)
{
    const long n_blocks = argc > 1 ? atol(argv[1]) : 2000000;
    const bool bench = argc > 2;  // bench-like regime: phaseAvg 50, xdelta 0.01, small offset, 37 dB, |y| < argv[2]
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    long n_ok[2] = {0, 0}, n_rej = 0;
    for (long it = 0; it < n_blocks; it++) {
        const int n = bench ? 50 : 2 + (int)(U(rng) * 200);
        const float xd = (float)(bench || U(rng) < 0.5 ? 0.01 : std::pow(10.0, -3 + 3 * U(rng)));
        const double y0 = bench ? atof(argv[2]) * (2 * U(rng) - 1)
                                : (U(rng) < 0.3 ? 1000.0 : U(rng) < 0.5 ? 30.0 : 1.0) * (2 * U(rng) - 1);
        const double slope = (bench || U(rng) < 0.5 ? 1e-3 : 0.1) * (2 * U(rng) - 1);
        const double sig = bench || U(rng) < 0.5 ? 0.04 : 0.4;
        std::normal_distribution<double> N(0.0, sig);
        static float yv[4096];
        // two blocks: the first one only brings the running sums into their steady rounding state
        for (int j = 0; j < n + 2 * B; j++) yv[j] = (float)(y0 + slope * j + N(rng));
        double ySum = 0, xySum = 0;
        for (int j = 0; j < n; j++) {
            ySum += (double)yv[j];
            float jx = (float)j * xd;
            xySum += (double)(jx * yv[j]);
        }
        double c[B], t[B];
        double ys = ySum;
        for (int blk = 0; blk < 2; blk++) {
            for (int j = 0; j < B; j++) {
                ys -= (double)yv[blk * B + j];
                c[j] = (double)xd * ys;
                ys += (double)yv[blk * B + n + j];
                float tt = yv[blk * B + n + j] * (float)(n - 1);
                tt = tt * xd;
                t[j] = (double)tt;
            }
            if (blk == 0) {
                double tmp[B];
                seq(xySum, c, t, B, tmp);
                xySum = tmp[B - 1];
            }
        }
        const int nvalid = U(rng) < 0.9 ? B : 1 + (int)(U(rng) * (B - 1));
        double ref[B];
        seq(xySum, c, t, nvalid, ref);
        Result g = grid(xySum, c, t, nvalid);
        bool same = true;
        for (int j = 0; j < nvalid; j++)
            if (bits(ref[j]) != bits(g.s[j]))
                same = false;
        if (same)
            n_ok[g.mode]++;
        else
            n_rej++;
    }
    printf("blocks %ld: candidates identical to the recurrence: mode A %ld, mode B %ld; different (-> chain) %ld = %.2f %%\n",
           n_blocks, n_ok[0], n_ok[1], n_rej, 100.0 * n_rej / n_blocks);
    return 0;
}
