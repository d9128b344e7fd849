/* smol_zones_label.h -- the labelling behind smol_hip_memory_zones() (smol_rt_zones.hip), apart from the device: windows 0 .. n-1 of a
 * region, a measurement `pair (a, b, &ms)' = the duration of two read streams side by side, one from the head of window a and one from the
 * head of window b (a == b: from the two halves of that window), and the rules that turn those durations into kinds. Kept in a header so
 * that tests/emu can run the rules against simulated devices (tests/test_zones_label.py). */

#pragma once

#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

namespace smolzones {

constexpr uint32_t PROBE_PASSES = 3, PROBE_RETRIES = 4;
constexpr float    ZONE_LINE = 1.03f, ZONE_DOUBT_LO = 1.02f, ZONE_DOUBT_HI = 1.04f, ZONE_GAP_MIN = 1.006f, ZONE_SURE = 0.988f;
constexpr uint32_t ZONE_KINDS_MAX = 8;
constexpr uint8_t  UNLABELLED = 0xff;

/* zones [0 .. n-1] receive the kinds (window 0 = kind 0, labels in order of first appearance), *kinds_out their number, *contrast_out
 * the slowest pair against two halves of one window. Returns 0, or what `pair' returned when that was not 0. `trace': every sweep's
 * times on stderr. */
template <class Pair>
int
label_windows (const uint64_t n, Pair &&pair, uint8_t *zones, uint32_t *kinds_out, float *contrast_out, const bool trace)
{
    int rc;
    float ms;
    std::fill (zones, zones + n, UNLABELLED);
    std::vector<float> t (n);
    uint32_t kinds = 0;
    float worst = 1.f;
    uint64_t ref_of [ZONE_KINDS_MAX] = { 0 };
    float same_of [ZONE_KINDS_MAX] = { 0.f };

    for (uint64_t ref = 0; ref < n; ref++)
    {
        if (zones [ref] != UNLABELLED)
            continue;
        if (kinds == ZONE_KINDS_MAX - 1)
        {
            /* windows that straddle a border read like nothing else: they share the last label */
            for (uint64_t w = ref; w < n; w++)
                if (zones [w] == UNLABELLED)
                    zones [w] = (uint8_t) kinds;
            kinds++;
            break;
        }
        zones [ref] = (uint8_t) kinds;
        ref_of [kinds] = ref;

        /* What two streams of ONE kind take: the two halves of a window, whatever its kind. Three windows are asked -- this one and the
         * first and last of the others --, once per pass like every other pair, and the middle answer counts: the halves of a region's
         * FIRST window read up to 4 % faster than any other pair of one kind (too fast a yardstick splits its kind), those of a window
         * that straddles a border 7 % slower (too slow a one merges two). */
        uint64_t cal [3] = { ref, ref, ref };
        uint32_t n_cal = 1;
        for (uint64_t w = ref + 1; w < n && n_cal < 2; w++)
            if (zones [w] == UNLABELLED)
                cal [n_cal++] = w;
        for (uint64_t w = n; w-- > ref + 1 && n_cal == 2;)
            if (zones [w] == UNLABELLED && w != cal [1])
                cal [n_cal++] = w;
        float halves [3] = { 1e30f, 1e30f, 1e30f };
        for (uint32_t pass = 0; pass < PROBE_PASSES; pass++)
        {
            for (uint32_t c = 0; c < n_cal; c++)
                for (uint32_t i = 0; i < 2; i++)
                {
                    if ((rc = pair (cal [c], cal [c], &ms)))
                        return rc;
                    halves [c] = std::min (halves [c], ms);
                }
            for (uint64_t w = ref + 1; w < n; w++)
            {
                if (zones [w] != UNLABELLED)
                    continue;
                if ((rc = pair (ref, w, &ms)))
                    return rc;
                t [w] = pass ? std::min (t [w], ms) : ms;           /* the quietest of the passes */
            }
        }
        std::sort (halves, halves + n_cal);
        float same = halves [n_cal / 2];            /* one window: its own; two: the slower (the first window's are the fast ones) */

        if (trace)
        {
            fprintf (stderr, "smol_hip_memory_zones: kind %u, first window %llu, its halves %.2f us; the others against that:", kinds,
                     (unsigned long long) ref, same * 1e3f);
            for (uint64_t w = ref + 1; w < n; w++)
                if (zones [w] == UNLABELLED)
                    fprintf (stderr, " %llu:%.3f", (unsigned long long) w, t [w] / same);
            fprintf (stderr, "\n");
        }

        /* Against that yardstick this window's own kind reads 0.985-1.025 and the other kinds 1.035-1.11 (the quietest of three
         * passes each): the line is drawn at 3 %, or in the middle of the widest gap that the sorted times show between 2 % and 4 %
         * (wider brackets find gaps INSIDE a sparse cluster of another kind). A window on the line is measured again. */
        float line = same * ZONE_LINE;
        {
            std::vector<float> v;
            for (uint64_t w = ref + 1; w < n; w++)
                if (zones [w] == UNLABELLED)
                    v.push_back (t [w]);
            std::sort (v.begin (), v.end ());
            float widest = ZONE_GAP_MIN;
            for (size_t i = 0; i + 1 < v.size (); i++)
            {
                const float mid = 0.5f * (v [i] + v [i + 1]);
                if (v [i + 1] / v [i] > widest && mid > same * ZONE_DOUBT_LO && mid < same * ZONE_DOUBT_HI)
                {
                    widest = v [i + 1] / v [i];
                    line = mid;
                }
            }
        }
        uint64_t other = ref;                       /* the member of this kind farthest from its first window */
        for (uint64_t w = ref + 1; w < n; w++)
            if (zones [w] == UNLABELLED)
            {
                for (uint32_t k = 0; k < PROBE_RETRIES && t [w] > line * 0.995f && t [w] < line * 1.005f; k++)
                {
                    if ((rc = pair (ref, w, &ms)))
                        return rc;
                    t [w] = std::min (t [w], ms);
                }
                worst = std::max (worst, t [w] / same);
                if (t [w] < line)
                {
                    zones [w] = (uint8_t) kinds;
                    if (t [w] < line * ZONE_SURE)
                        other = w;
                }
            }
        /* ... and a window that came in just under the line is asked again beside that member: the slow end of this kind and the
         * fast end of another are a per cent apart (2 processes in 24 had one to three windows of another kind among this one's) */
        if (other != ref)
            for (uint64_t w = ref + 1; w < n; w++)
                if (zones [w] == (uint8_t) kinds && w != other && t [w] >= line * ZONE_SURE)
                {
                    float best = 1e30f;
                    for (uint32_t k = 0; k < 2; k++)
                    {
                        if ((rc = pair (other, w, &ms)))
                            return rc;
                        best = std::min (best, ms);
                    }
                    if (best >= line)
                        zones [w] = UNLABELLED;
                }
        /* A fast pair is proof of one kind -- nothing makes two streams from two kinds fast --, a slow pair is only a hint: a
         * disturbed measurement is slow too. Every window that read slowly beside the kind's first window is asked again beside
         * another of its members before it is left for the next kind. */
        if (other != ref)
            for (uint64_t w = ref + 1; w < n; w++)
                if (zones [w] == UNLABELLED)
                    for (uint32_t k = 0; k < 2; k++)
                    {
                        if ((rc = pair (other, w, &ms)))
                            return rc;
                        if (ms < line)
                        {
                            zones [w] = (uint8_t) kinds;
                            break;
                        }
                    }
        same = line / ZONE_LINE;            /* for the second look below */
        same_of [kinds] = same;
        kinds++;
    }

    /* a kind of a few windows is a window that was disturbed during every one of its measurements, or one that straddles a
     * border: ask again, at leisure, beside the first window of every populous kind */
    std::vector<uint64_t> count (kinds, 0);
    for (uint64_t w = 0; w < n; w++)
        count [zones [w]]++;
    const uint64_t few = std::max<uint64_t> (1, n / 8);
    for (uint64_t w = 0; w < n; w++)
    {
        if (count [zones [w]] > few)
            continue;
        for (uint32_t k = 0; k < kinds; k++)
        {
            if (count [k] <= few)
                continue;
            float best = 1e30f;
            for (uint32_t i = 0; i < 2 * PROBE_RETRIES; i++)
            {
                if ((rc = pair (ref_of [k], w, &ms)))
                    return rc;
                best = std::min (best, ms);
            }
            if (best < same_of [k] * ZONE_LINE)
            {
                count [zones [w]]--;
                zones [w] = (uint8_t) k;
                count [k]++;
                break;
            }
        }
    }
    /* labels in use, renumbered in order of first appearance */
    uint8_t renum [ZONE_KINDS_MAX];
    std::fill (renum, renum + ZONE_KINDS_MAX, UNLABELLED);
    uint32_t used = 0;
    for (uint64_t w = 0; w < n; w++)
    {
        if (renum [zones [w]] == UNLABELLED)
            renum [zones [w]] = (uint8_t) used++;
        zones [w] = renum [zones [w]];
    }
    kinds = used;
    if (kinds_out)
        *kinds_out = kinds;
    if (contrast_out)
        *contrast_out = worst;
    return 0;
}

} /* namespace smolzones */
