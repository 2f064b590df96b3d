// pdt_timeaxis.h -- closed form of the reference's running-sum time axis (host side).
//
// The reference stamps sample i with time_{i} = fl(time_{i-1} + Ts), a float32 (POES) or
// float64 (ARGOS) accumulator that is never reset (common/wave.c:91,96-97,167-168;
// SURVEY Appendix B Q1).  Time stamps are only ever *printed* for the handful of bits that
// complete a sync word, so instead of materialising one value per sample the GPU pipeline
// carries sample indices and this class evaluates T(m) = value after m additions on demand.
//
// Inside one binade [2^e, 2^(e+1)) every accumulator value is a multiple of the binade's ulp
// u, so fl(t + Ts) = t + RN_u(Ts): the increment is a constant multiple of u (after at most
// one step in the round-half-even tie case).  The table therefore stores, per binade, the
// first value reached with real floating-point additions and the constant increment; the
// steps that cross a binade boundary are always taken with a real addition.
#pragma once
#include <math.h>
#include <stdint.h>

#include <vector>

namespace pdt {

template <typename F> class TimeAxis {
  public:
    void init(F ts)
    {
        Ts = ts;
        segs.clear();
        cur_m = 0;
        cur_t = 0;
        stalled = false;
    }

    // value of the accumulator after m additions (T(0) = 0)
    F at(uint64_t m)
    {
        while (!stalled && cur_m < m) extend();
        // find the segment containing m (segments are in increasing m0 order)
        size_t lo = 0, hi = segs.size();
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (segs[mid].m0 <= m) lo = mid; else hi = mid;
        }
        if (segs.empty() || m < segs[0].m0) return 0;
        const Seg &s = segs[lo];
        uint64_t k = m - s.m0;
        if (k > s.count) k = s.count;                 // only possible in the final (stalled) segment
        // exact: t0 and d are multiples of the binade ulp and the result stays inside the binade
        return from_units(s, k);
    }

  private:
    struct Seg {
        uint64_t m0;      // additions done at the segment's first value
        F t0;             // that value
        F d;              // constant increment
        uint64_t count;   // further additions covered: values t0 + k*d, k = 0..count
    };
    F Ts = 0;
    std::vector<Seg> segs;
    uint64_t cur_m = 0;   // additions covered so far
    F cur_t = 0;
    bool stalled = false;

    static F from_units(const Seg &s, uint64_t k)
    {
        if (k == 0 || s.d == 0) return s.t0;
        int e;
        (void)frexp((double)s.t0, &e);                               // t0 in [2^(e-1), 2^e)
        const int digits = (sizeof(F) == 4) ? 24 : 53;
        const double u = ldexp(1.0, e - digits);                     // ulp of the binade
        const unsigned __int128 mt = (unsigned __int128)(uint64_t)llround((double)s.t0 / u);
        const unsigned __int128 md = (unsigned __int128)(uint64_t)llround((double)s.d / u);
        const unsigned __int128 v = mt + md * (unsigned __int128)k;  // < 2^digits by construction
        return (F)ldexp((double)(uint64_t)v, e - digits);
    }

    void push_single(uint64_t m0, F t0)
    {
        Seg s;
        s.m0 = m0;
        s.t0 = t0;
        s.d = 0;
        s.count = 0;
        segs.push_back(s);
    }

    static int binade(F v)
    {
        int e;
        (void)frexp((double)v, &e);
        return e;
    }

    // advance the table by at least one addition
    void extend()
    {
        // real additions until two consecutive increments inside one binade agree
        volatile F t1 = (F)(cur_t + Ts);
        push_single(cur_m + 1, t1);
        cur_m += 1;
        if (t1 == cur_t) {                      // stall: Ts below half an ulp
            segs.back().count = UINT64_MAX / 2;
            stalled = true;
            cur_t = t1;
            return;
        }
        const F prev = cur_t;
        cur_t = t1;
        volatile F t2 = (F)(cur_t + Ts);
        if (t2 == cur_t) return;                // next call detects the stall
        const F d1 = (F)(t1 - prev), d2 = (F)(t2 - t1);
        if (prev == 0 || binade(prev) != binade(t1) || binade(t1) != binade(t2) || d1 != d2) return;
        // constant increment d2 from t1 on, while the value stays inside t1's binade
        int e;
        (void)frexp((double)t1, &e);
        const int digits = (sizeof(F) == 4) ? 24 : 53;
        const double u = ldexp(1.0, e - digits);
        const uint64_t mt = (uint64_t)llround((double)t1 / u);
        const uint64_t md = (uint64_t)llround((double)d2 / u);
        const uint64_t top = 1ull << digits;    // first mantissa value of the next binade
        if (md == 0) return;
        const uint64_t k = (top - 1 - mt) / md; // largest k with mt + k*md <= top-1
        Seg &s = segs.back();
        s.d = d2;
        s.count = k;
        cur_m += k;
        cur_t = from_units(s, k);
    }
};

}  // namespace pdt
