// rt_team.inl - pipeline 4, included by rt_kernels.hip (once per flavour): path waves and walker waves.
//
// In the lock-step kernels the 64 rays of a wave walk a BVH parent together and the wave waits for its
// slowest lane: with four or five of 64 lanes still busy on average, about half of the kernel's issue
// slots go to walks that serve a handful of rays. Here a workgroup of 16 waves fills one compute unit
// and its waves take two roles:
//   path waves    (12 by default) own 8x8 pixel tiles and run everything of the lock-step kernel but
//                 the BVH walks: camera rays, the wave-uniform object loop with the leaf parents (walls)
//                 tested in place, shading, accumulation. When a ray reaches a run of BVH parents
//                 the lane writes a job - ray, closest hit so far, the parents whose boxes it may enter -
//                 into its own slot in LDS, the wave appends the slot numbers to a ring in LDS with
//                 one atomic (ballot + popcount give every lane its place) and sleeps until its jobs
//                 are done;
//   walker waves  (4 by default, one per SIMD) never shade. Every lane takes the next job from the ring
//                 as soon as it is free, whichever path wave it came from, walks the listed parents in
//                 order - parent box, two-child records, triangles, one step per iteration - and
//                 writes the hit back into the job's slot. The walks of all twelve path waves keep the
//                 walker lanes full, so a walk step is issued for ~40 rays rather than ~5.
// The queue never leaves the compute unit: jobs, ring, completion counters and the walkers' traversal
// stacks are all in LDS (about 90 KB per workgroup); nothing but LDS atomics and workgroup-scope
// fences orders producers and consumers, and a workgroup depends on no other workgroup.
//
// Every ray still meets objects, records and triangles in the order its traversal flavour prescribes
// (reference order: raytrace.comp:458-610; near-first inside rebuilt trees), with the closest hit
// carried from parent to parent, so the images are those of the lock-step kernels bit for bit.

#define RT_TEAM_WAVES 16
#define RT_TEAM_RING 2048            /* ring entries; at most (16 - walkers) * 2 * 64 jobs are outstanding */
#define RT_TEAM_IN 10                /* dwords per job, in: origin 0-2, direction 3-5, closest 6, meta 7, mask 8-9 */
#define RT_TEAM_OUT 6                /* dwords per job, out: closest, tLocal, u, v, object, triangle */
#define RT_TEAM_SLOTS 2              /* jobs a lane can have in flight: a regular ray and a shadow ray */
#define RT_TEAM_EMPTY (-1)
#define RT_TEAM_NEXT 0x7ffffffd      /* walker lane: between two parents of its job */
#define RT_TEAM_ANYHIT 0x80000000u   /* job meta: stop at the first accepted hit */
#define RT_TEAM_PERMUTE 0x40000000u  /* job meta: positions map to objects through DevScene::anyOrder */

namespace {

// Workgroup-shared state, carved out of dynamic LDS.
struct TeamShared {
    unsigned* ctrl;       // [0] ring head (consumers), [1] ring tail (producers), [2] path waves finished
    int* ring;            // RT_TEAM_RING job numbers or RT_TEAM_EMPTY
    unsigned* pending;    // per path wave: jobs of its current batch not yet done
    float* in;            // [path wave][slot][RT_TEAM_IN][64]; a job's number is (path wave * RT_TEAM_SLOTS + slot) * 64 + lane
    float* out;           // [path wave][slot][RT_TEAM_OUT][64]
    uint2* stacks;        // [walker wave][depth][64]
};

__device__ __forceinline__ TeamShared team_shared(void* base, int pathWaves, int walkers, int stackDepth)
{
    TeamShared t;
    char* p = static_cast<char*>(base);
    t.ctrl = reinterpret_cast<unsigned*>(p); p += 64;
    t.ring = reinterpret_cast<int*>(p); p += RT_TEAM_RING * sizeof(int);
    t.pending = reinterpret_cast<unsigned*>(p); p += 64;
    t.in = reinterpret_cast<float*>(p); p += (size_t)pathWaves * RT_TEAM_SLOTS * RT_TEAM_IN * 64 * sizeof(float);
    t.out = reinterpret_cast<float*>(p); p += (size_t)pathWaves * RT_TEAM_SLOTS * RT_TEAM_OUT * 64 * sizeof(float);
    t.stacks = reinterpret_cast<uint2*>(p);
    (void)walkers; (void)stackDepth;
    return t;
}

__device__ __forceinline__ unsigned team_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int team_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// What a path wave knows about itself.
struct TeamPath {
    TeamShared sh;
    int wave;             // index among the path waves of the workgroup
    int lane;
};

// Hands the jobs of the lanes in `emit` (slot k of each lane) to the walkers. Whole wave.
__device__ __forceinline__ void team_emit(const TeamPath& tp, int k, bool emit, V3 org, V3 dirN, float closest, unsigned long long mask, unsigned meta)
{
    const unsigned long long m = __ballot(emit);
    if (m == 0) return;
    const unsigned n = (unsigned)__popcll(m);
    const int first = (tp.wave * RT_TEAM_SLOTS + k) * 64;
    float* in = tp.sh.in + (size_t)(tp.wave * RT_TEAM_SLOTS + k) * RT_TEAM_IN * 64 + tp.lane;
    if (emit) {
        in[0 * 64] = org.x; in[1 * 64] = org.y; in[2 * 64] = org.z;
        in[3 * 64] = dirN.x; in[4 * 64] = dirN.y; in[5 * 64] = dirN.z;
        in[6 * 64] = closest;
        in[7 * 64] = __uint_as_float(meta);
        in[8 * 64] = __uint_as_float((unsigned)mask);
        in[9 * 64] = __uint_as_float((unsigned)(mask >> 32));
    }
    unsigned base = 0;
    if (tp.lane == 0) {
        __hip_atomic_fetch_add(&tp.sh.pending[tp.wave], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = __hip_atomic_fetch_add(&tp.sh.ctrl[1], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // jobs and counter before their ring entries
    base = (unsigned)__shfl((int)base, 0, 64);
    if (emit) {
        const unsigned rank = (unsigned)__popcll(m & ((1ull << tp.lane) - 1ull));
        int* entry = &tp.sh.ring[(base + rank) & (RT_TEAM_RING - 1)];
        // The entry of one lap ago was claimed long since (fewer jobs are in flight than the ring has
        // entries, and claims go in ring order); its walker may be about to clear it, though.
        while (__hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != RT_TEAM_EMPTY) __builtin_amdgcn_s_sleep(1);
        __hip_atomic_store(entry, first + tp.lane, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Sleeps until every job this wave has emitted is done. Whole wave.
__device__ __forceinline__ void team_wait(const TeamPath& tp)
{
#ifdef RT_SECTION_TIMERS
    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
    while (team_load(&tp.sh.pending[tp.wave]) != 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef RT_SECTION_TIMERS
    if (tp.lane == 0) atomicAdd(&g_sectionCycles[4], __builtin_amdgcn_s_memtime() - w0);
#endif
}

// The answer to this lane's job in slot k: true, with `best` filled, when the walk accepted a hit.
__device__ __forceinline__ bool team_result(const TeamPath& tp, int k, Best& best)
{
    const float* out = tp.sh.out + (size_t)(tp.wave * RT_TEAM_SLOTS + k) * RT_TEAM_OUT * 64 + tp.lane;
    const int object = __float_as_int(out[4 * 64]);
    if (object < 0) return false;
    best.closest = out[0 * 64]; best.tLocal = out[1 * 64]; best.u = out[2 * 64]; best.v = out[3 * 64];
    best.object = object; best.tri = __float_as_int(out[5 * 64]);
    return true;
}

// Emit, wait, take the answer: the walk of a run of BVH parents in the middle of an ordered object loop.
__device__ __forceinline__ void team_walk(const TeamPath& tp, bool emit, V3 org, V3 dirN, unsigned long long mask, unsigned meta, Best& best)
{
    if (!__any(emit)) return;
    team_emit(tp, 0, emit, org, dirN, best.closest, mask, meta);
    team_wait(tp);
    if (emit) (void)team_result(tp, 0, best);
}

// GetHit (raytrace.comp:448-610) as a path wave runs it: trace() with the walks of the BVH parents
// handed to the walker waves, one job per lane and run of consecutive BVH parents. `on`: the lane has
// a ray (every lane of the wave goes through the loop, so that the wave-wide steps stay wave-wide).
template <bool ANYHIT>
__device__ __forceinline__ void trace_team(const DevScene& S, int parentCount, V3 org, V3 dirN, bool on, float tMin, float tMax, Best& best,
                                           const TeamPath& tp)
{
    best.closest = tMax;
    best.object = -1;
    best.tri = 0; best.u = 0.0f; best.v = 0.0f; best.tLocal = 0.0f;
    V3 winv;
    const bool parW = lean_inverse(dirN, winv) && on;
    const bool permute = ANYHIT && parentCount == S.objectCount;
    const bool tightScene = S.nearFirstBase != 0x7fffffff;
    bool live = on;            // ANYHIT: false once something was hit
    int pos = 0;
    while (pos < parentCount) {
        if (ANYHIT && !__any(live)) break;
        const int i = permute ? S.anyOrder[pos] : pos;
        const DevObject* ob = S.objects + i;   // wave-uniform address
        const int root = ob->root;             // wave-uniform
        if (root >= 0) {
            // a run of BVH parents: which of their boxes can this ray enter at all (slab test against the
            // closest hit of this moment; the walker repeats the reference's full test, parent by parent,
            // against the closest hit it has by then)
            const int runStart = pos;
            unsigned long long mask = 0;
            int n = 0;
            while (pos < parentCount && n < 64) {
                const int j = permute ? S.anyOrder[pos] : pos;
                const DevObject* o2 = S.objects + j;
                if (o2->root < 0) break;
                float E;
                const V3 bmn = ld3(o2->bmin), bmx = ld3(o2->bmax);
                const bool ok = live && (__any(parW) ? lean_box<true>(org, winv, bmn, bmx, tMin, E) : lean_box<false>(org, winv, bmn, bmx, tMin, E)) &&
                                (best.closest > E);
                if (ok) mask |= 1ull << n;
                ++pos; ++n;
            }
            const unsigned meta = (unsigned)runStart | (ANYHIT ? RT_TEAM_ANYHIT : 0u) | (permute ? RT_TEAM_PERMUTE : 0u);
            team_walk(tp, mask != 0, org, dirN, mask, meta, best);
            if (ANYHIT && best.object >= 0) live = false;
            continue;
        }
        ++pos;
        const V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
        float E;
        bool ok = live && (__any(parW) ? lean_box<true>(org, winv, bmn, bmx, tMin, E) : lean_box<false>(org, winv, bmn, bmx, tMin, E)) &&
                  (best.closest > E);
        if (ok && !tightScene) ok = !(distance_to_aabb(org, bmn, bmx) > best.closest);
        // a parent that is one leaf (a wall): records through the scalar cache, once for the wave
        if (!(root > RT_REF_LIVE_MIN) || !__any(ok)) continue;
        const V3 lo = xform(ob->inv, org, 1.0f);
        const V3 ld = normalize3(xform(ob->inv, dirN, 0.0f));
        const float4* rec = reinterpret_cast<const float4*>(S.tris + ~root);
        for (;;) {
            const float4 t0 = rec[0], t1 = rec[1], t2 = rec[2];
            if (ok && triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, i, best)) {
                if (ANYHIT) { ok = false; live = false; }
            }
            if (__float_as_int(t0.w) <= 1) break;
            rec += 3;
        }
    }
}

// RayColour (raytrace.comp:692-769) on a path wave. The wave stays together from GetHit call to GetHit
// call; the scatter is done before the shadow ray is traced, with the sun term of the hit kept aside
// until the shadow ray's answer is in (same values, same order of additions to `result`).
__device__ __forceinline__ V3 ray_colour_team(const DevScene& S, const rt_push_constants& pc, Ray ray, bool alive, uint32_t seed, V3 sunDirN,
                                              const TeamPath& tp)
{
    V3 result = mk(0.0f, 0.0f, 0.0f);
    V3 throughput = mk(1.0f, 1.0f, 1.0f);
    const int bounces = pc.maxBounces + 1;
    for (int bounce = 0; bounce < bounces; ++bounce) {
        if (!__any(alive)) break;
        const V3 dirN = normalize3(ray.d);
        Best best;
        trace_team<false>(S, pc.parentBVHCount, ray.o, dirN, alive, 0.001f, 1e20f, best, tp);
        bool shadow = false;
        V3 so = mk(0.0f, 0.0f, 0.0f), sunTerm = mk(0.0f, 0.0f, 0.0f);
        if (alive) {
            HitRec rec;
            finish_hit(S, ray.o, dirN, best, rec);
            if (!rec.hit) {
                result = result + throughput * environment_colour(pc, ray.d);
                alive = false;
            } else if (!rec.frontFace) {
                alive = false;
            } else {
                const DevMaterial m = load_material(S, rec.matIndex);
                const V3 emitted = ld3(m.colour) * m.emission;
                result = result + throughput * emitted;
                if (m.emission > 0.0f) {
                    alive = false;
                } else {
                    const V3 negSun = -ld3(pc.sunDirection);
                    const float nDotL = gmax(dot3(rec.normal, negSun), 0.0f);
                    if (nDotL > 0.0f) {
                        shadow = true;
                        so = rec.point + rec.normal * 0.01f;
                        const V3 sun = ld3(pc.sunColour) * pc.sunIntensity;
                        const float INV_PI = 0.31830988618f;
                        const V3 direct = sun * (nDotL * INV_PI);
                        if (m.refractionIndex == 0.0f) sunTerm = (throughput * ld3(m.colour)) * direct;
                        else sunTerm = (throughput * reflectance(nDotL, m.refractionIndex)) * direct;
                    }
                    V3 attenuation;
                    Ray scattered;
                    material_scatter(m, ray.d, rec, seed, attenuation, scattered);
                    throughput = throughput * attenuation;
                    ray = scattered;
                }
            }
        }
        if (__any(shadow)) {
            Best sb;
            trace_team<true>(S, pc.parentBVHCount, so, sunDirN, shadow, 0.001f, 1e20f, sb, tp);
            if (shadow && sb.object < 0) result = result + sunTerm;
        }
    }
    return result;
}

// RayColour for scenes whose GetHit does not depend on the visiting order (DevScene::orderFree). Per
// bounce the wave hands BOTH rays it has to trace - the shadow ray of the hit just shaded and the
// scattered ray - to the walkers first (every BVH parent whose box a ray can enter, in one job per ray),
// tests the leaf parents itself while the walkers work, then waits once and merges by distance. Both groups
// are gone through in the any-hit order (leaf parents, BVH parents, each sorted towards the sun), so two
// parents that are hit at exactly the same distance are decided by that order inside a group and by the
// object index between the groups - not always as the scene-order loop decides them: such ties are the
// near-first flavour's stated tolerance, and the only case in which this kernel's image can differ from
// the lock-step kernels' near-first image. The sun term of a hit is
// added when its shadow ray's answer is in, one iteration later, still before anything the next hit
// adds: the additions to `result` keep the reference's order.
__device__ __forceinline__ V3 ray_colour_team_async(const DevScene& S, const rt_push_constants& pc, Ray ray, bool alive, uint32_t seed,
                                                    V3 sunDirN, const TeamPath& tp)
{
    V3 result = mk(0.0f, 0.0f, 0.0f);
    V3 throughput = mk(1.0f, 1.0f, 1.0f);
    const int bounces = pc.maxBounces + 1;
    const int P = S.objectCount, leafCount = S.anyLeafCount;
    const float tMin = 0.001f;
    const unsigned metaBase = (unsigned)leafCount | RT_TEAM_PERMUTE;
    V3 sunInv;
    const bool sunPar = lean_inverse(sunDirN, sunInv);
    bool shadow = false;
    V3 so = mk(0.0f, 0.0f, 0.0f), sunTerm = mk(0.0f, 0.0f, 0.0f);
    for (int bounce = 0; ; ++bounce) {
        const bool want = alive && bounce < bounces;
        if (!__any(want || shadow)) break;
        const V3 dirN = normalize3(ray.d);
        V3 winv;
        const bool parW = lean_inverse(dirN, winv) && want;
        const bool anyParR = __any(parW), anyParS = sunPar && __any(shadow);

        // which BVH parents (positions leafCount.. of the any-hit order) each of the two rays can enter
        unsigned long long maskR = 0, maskS = 0;
        for (int pos = leafCount; pos < P; ++pos) {
            const DevObject* ob = S.objects + S.anyOrder[pos];   // wave-uniform
            const V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
            float E;
            if (want && (anyParR ? lean_box<true>(ray.o, winv, bmn, bmx, tMin, E) : lean_box<false>(ray.o, winv, bmn, bmx, tMin, E)))
                maskR |= 1ull << (pos - leafCount);
            if (shadow && (anyParS ? lean_box<true>(so, sunInv, bmn, bmx, tMin, E) : lean_box<false>(so, sunInv, bmn, bmx, tMin, E)))
                maskS |= 1ull << (pos - leafCount);
        }
        const bool jobR = want && maskR != 0, jobS = shadow && maskS != 0;
        const bool anyJob = __any(jobR || jobS);
        team_emit(tp, 0, jobR, ray.o, dirN, 1e20f, maskR, metaBase);
        team_emit(tp, 1, jobS, so, sunDirN, 1e20f, maskS, metaBase | RT_TEAM_ANYHIT);

        // the leaf parents, here, while the walkers work: closest hit for the regular ray, any hit for the shadow ray
        Best best;
        best_reset(best, 1e20f);
        bool blocked = false;
        for (int pos = 0; pos < leafCount; ++pos) {
            const int i = S.anyOrder[pos];
            const DevObject* ob = S.objects + i;   // wave-uniform
            const int root = ob->root;
            if (!(root > RT_REF_LIVE_MIN)) continue;
            const V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
            float E;
            bool okR = want && (anyParR ? lean_box<true>(ray.o, winv, bmn, bmx, tMin, E) : lean_box<false>(ray.o, winv, bmn, bmx, tMin, E)) &&
                       (best.closest > E);
            bool okS = shadow && !blocked && (anyParS ? lean_box<true>(so, sunInv, bmn, bmx, tMin, E) : lean_box<false>(so, sunInv, bmn, bmx, tMin, E));
            if (!__any(okR || okS)) continue;
            const V3 loR = xform(ob->inv, ray.o, 1.0f), ldR = normalize3(xform(ob->inv, dirN, 0.0f));
            const V3 loS = xform(ob->inv, so, 1.0f), ldS = normalize3(xform(ob->inv, sunDirN, 0.0f));
            const float4* rec = reinterpret_cast<const float4*>(S.tris + ~root);
            for (;;) {
                const float4 t0 = rec[0], t1 = rec[1], t2 = rec[2];
                if (okR) (void)triangle_hit(t0, t1, t2, loR, ldR, ob->fwd, ray.o, dirN, tMin, i, best);
                if (okS) {
                    Best sb;
                    best_reset(sb, 1e20f);
                    if (triangle_hit(t0, t1, t2, loS, ldS, ob->fwd, so, sunDirN, tMin, i, sb)) { blocked = true; okS = false; }
                }
                if (__float_as_int(t0.w) <= 1) break;
                rec += 3;
            }
        }

        if (anyJob) team_wait(tp);
        if (jobR) {
            Best w;
            if (team_result(tp, 0, w) && (w.closest < best.closest || (w.closest == best.closest && w.object < best.object))) best = w;
        }
        if (jobS && !blocked) {
            Best w;
            if (team_result(tp, 1, w)) blocked = true;
        }
        if (shadow && !blocked) result = result + sunTerm;
        shadow = false;

        if (want) {
            HitRec rec;
            finish_hit(S, ray.o, dirN, best, rec);
            if (!rec.hit) {
                result = result + throughput * environment_colour(pc, ray.d);
                alive = false;
            } else if (!rec.frontFace) {
                alive = false;
            } else {
                const DevMaterial m = load_material(S, rec.matIndex);
                const V3 emitted = ld3(m.colour) * m.emission;
                result = result + throughput * emitted;
                if (m.emission > 0.0f) {
                    alive = false;
                } else {
                    const V3 negSun = -ld3(pc.sunDirection);
                    const float nDotL = gmax(dot3(rec.normal, negSun), 0.0f);
                    if (nDotL > 0.0f) {
                        shadow = true;
                        so = rec.point + rec.normal * 0.01f;
                        const V3 sun = ld3(pc.sunColour) * pc.sunIntensity;
                        const float INV_PI = 0.31830988618f;
                        const V3 direct = sun * (nDotL * INV_PI);
                        if (m.refractionIndex == 0.0f) sunTerm = (throughput * ld3(m.colour)) * direct;
                        else sunTerm = (throughput * reflectance(nDotL, m.refractionIndex)) * direct;
                    }
                    V3 attenuation;
                    Ray scattered;
                    material_scatter(m, ray.d, rec, seed, attenuation, scattered);
                    throughput = throughput * attenuation;
                    ray = scattered;
                }
            }
        }
    }
    return result;
}

// A walker wave: every lane takes jobs from the ring and walks them, one step per iteration.
__device__ __forceinline__ void team_walker(const DevScene& S, const TeamShared& sh, int pathWaves, uint2* stack, int lane, int refillMin)
{
    const float tMin = 0.001f;
    int slot = -1;             // the job this lane works on (path wave * 64 + lane), -1: none
    unsigned meta = 0;
    unsigned long long mask = 0;
    V3 winv = mk(1.0f, 1.0f, 1.0f);
    bool parW = false, parL = false, nearFirst = false;
    LaneTrace t;
    t.rayO = mk(0.0f, 0.0f, 0.0f); t.dirN = mk(0.0f, 0.0f, 1.0f); t.lo = t.rayO; t.ld = t.dirN; t.linv = winv;
    t.best.closest = 1e20f; t.best.tLocal = t.best.u = t.best.v = 0.0f; t.best.object = -1; t.best.tri = 0;
    t.obj = 0; t.cur = RT_TEAM_NEXT; t.sp = 0;
    const bool tightScene = S.nearFirstBase != 0x7fffffff;

#ifdef RT_SECTION_TIMERS
    unsigned long long dIter = 0, dLanes = 0, dIdle = 0, dJobs = 0;
    const unsigned long long dStart = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        // ---- refill, only when enough lanes are free (or nothing is being walked): the wave claims ring
        // positions that the path waves have already announced (never ahead of the tail: a position
        // claimed early and read late could be overwritten a lap later), at most one per free lane ----
        const int walking = __popcll(__ballot(slot >= 0));
        if (walking == 0 || walking <= 64 - refillMin) {
            const unsigned long long want = __ballot(slot < 0);
            unsigned base = 0, got = 0;
            if (lane == 0) {
                const unsigned n = (unsigned)__popcll(want);
                for (;;) {
                    unsigned h = __hip_atomic_load(&sh.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned tl = __hip_atomic_load(&sh.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned avail = tl - h;     // the tail is read after the head and both only grow
                    got = n < avail ? n : avail;
                    if (got == 0) break;
                    if (__hip_atomic_compare_exchange_strong(&sh.ctrl[0], &h, h + got, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        base = h;
                        break;
                    }
                }
            }
            base = (unsigned)__shfl((int)base, 0, 64);
            got = (unsigned)__shfl((int)got, 0, 64);
            if (slot < 0 && (unsigned)__popcll(want & ((1ull << lane) - 1ull)) < got) {
                int* entry = &sh.ring[(base + (unsigned)__popcll(want & ((1ull << lane) - 1ull))) & (RT_TEAM_RING - 1)];
                // announced, so its path wave is writing it at this moment if it is not there yet
                int v = __hip_atomic_load(entry, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (v < 0) {
                    __builtin_amdgcn_s_sleep(1);
                    v = __hip_atomic_load(entry, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __hip_atomic_store(entry, RT_TEAM_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                slot = v;
                const float* in = sh.in + (size_t)(v >> 6) * RT_TEAM_IN * 64 + (v & 63);   // v >> 6 = path wave * RT_TEAM_SLOTS + slot
                t.rayO = mk(in[0 * 64], in[1 * 64], in[2 * 64]);
                t.dirN = mk(in[3 * 64], in[4 * 64], in[5 * 64]);
                t.best.closest = in[6 * 64];
                t.best.object = -1; t.best.tri = 0; t.best.tLocal = 0.0f; t.best.u = 0.0f; t.best.v = 0.0f;
                meta = __float_as_uint(in[7 * 64]);
                mask = (unsigned long long)__float_as_uint(in[8 * 64]) | ((unsigned long long)__float_as_uint(in[9 * 64]) << 32);
                parW = lean_inverse(t.dirN, winv);
                t.cur = RT_TEAM_NEXT;
            }
#ifdef RT_SECTION_TIMERS
            dJobs += __popcll(__ballot(slot >= 0)) - walking;
#endif
            if (!__any(slot >= 0)) {
#ifdef RT_SECTION_TIMERS
                dIdle++;
#endif
                // nothing to do: leave when every path wave of the workgroup has finished (no job is outstanding then)
                if (team_load(&sh.ctrl[2]) >= (unsigned)pathWaves) break;
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
        }

#ifdef RT_SECTION_TIMERS
        dIter++; dLanes += __popcll(__ballot(slot >= 0));
#endif
        if (slot >= 0) {
            // between two parents: the job is done, or its next parent is due
            if (t.cur == RT_TEAM_NEXT) {
                const bool settled = (meta & RT_TEAM_ANYHIT) && t.best.object >= 0;
                if (mask == 0 || settled) {
                    float* out = sh.out + (size_t)(slot >> 6) * RT_TEAM_OUT * 64 + (slot & 63);
                    out[0 * 64] = t.best.closest; out[1 * 64] = t.best.tLocal; out[2 * 64] = t.best.u; out[3 * 64] = t.best.v;
                    out[4 * 64] = __int_as_float(t.best.object); out[5 * 64] = __int_as_float(t.best.tri);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __hip_atomic_fetch_sub(&sh.pending[(slot >> 6) / RT_TEAM_SLOTS], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    slot = -1;
                } else {
                    const int bit = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const int pos = (int)(meta & 0x00ffffffu) + bit;
                    t.obj = (meta & RT_TEAM_PERMUTE) ? S.anyOrder[pos] : pos;
                }
            }
        }
        if (slot >= 0) {
            // Whatever the lane is at - a parent (DevObject), a two-child record (DevNode) or a triangle
            // (DevTri) - the first 64 bytes of that record are fetched here, before the three kinds of step
            // part ways, so that an iteration waits for memory once.
            const bool atBox = t.cur == RT_TEAM_NEXT;
            const float4* rec = atBox ? reinterpret_cast<const float4*>(S.objects + t.obj)
                                      : (t.cur >= 0 ? reinterpret_cast<const float4*>(S.nodes + t.cur)
                                                    : reinterpret_cast<const float4*>(S.tris + ~t.cur));
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
            int next;
            if (atBox) {
                // parent box, raytrace.comp:458-486
                const V3 bmn = mk(r0.x, r0.y, r0.z), bmx = mk(r1.x, r1.y, r1.z);
                float E;
                bool ok = (__any(parW) ? lean_box<true>(t.rayO, winv, bmn, bmx, tMin, E) : lean_box<false>(t.rayO, winv, bmn, bmx, tMin, E)) &&
                          (t.best.closest > E);
                if (ok && !tightScene) ok = !(distance_to_aabb(t.rayO, bmn, bmx) > t.best.closest);
                next = RT_TEAM_NEXT;
                if (ok) {
                    const float4 i0 = r2, i1 = r3, i2 = rec[4];
                    t.lo = mk((i0.x * t.rayO.x + i0.y * t.rayO.y) + (i0.z * t.rayO.z + i0.w * 1.0f),
                              (i1.x * t.rayO.x + i1.y * t.rayO.y) + (i1.z * t.rayO.z + i1.w * 1.0f),
                              (i2.x * t.rayO.x + i2.y * t.rayO.y) + (i2.z * t.rayO.z + i2.w * 1.0f));
                    t.ld = normalize3(mk((i0.x * t.dirN.x + i0.y * t.dirN.y) + (i0.z * t.dirN.z + i0.w * 0.0f),
                                         (i1.x * t.dirN.x + i1.y * t.dirN.y) + (i1.z * t.dirN.z + i1.w * 0.0f),
                                         (i2.x * t.dirN.x + i2.y * t.dirN.y) + (i2.z * t.dirN.z + i2.w * 0.0f)));
                    parL = lean_inverse(t.ld, t.linv);
                    t.sp = 0;
                    next = __float_as_int(r0.w);      // root: a two-child record (a run holds BVH parents only)
                    nearFirst = next >= S.nearFirstBase;
                }
            } else if (t.cur >= 0) {
                NodeVisit nv;
                const int2 recRefs = make_int2(__float_as_int(r3.x), __float_as_int(r3.y));
                if (__any(parL)) node_boxes_loaded<true>(r0, r1, r2, recRefs, t.lo, t.linv, tMin, nv);
                else node_boxes_loaded<false>(r0, r1, r2, recRefs, t.lo, t.linv, tMin, nv);
                int2 refs = nv.refs;
                const bool goF = refs.x > RT_REF_LIVE_MIN && nv.okF && (t.best.closest > nv.EF);
                bool canS = refs.y > RT_REF_LIVE_MIN && nv.okS;
                if (nearFirst) {
                    // rebuilt tree: nearer child first, a child behind the closest hit is dropped at once
                    canS = canS && (t.best.closest > nv.ES);
                    if (goF && canS && nv.ES < nv.EF) {
                        const int r = refs.x; refs.x = refs.y; refs.y = r;
                        const float e = nv.EF; nv.EF = nv.ES; nv.ES = e;
                    }
                }
                if (goF) {
                    if (canS) { stack[t.sp * 64] = make_uint2((uint32_t)refs.y, __float_as_uint(nv.ES)); ++t.sp; }
                    next = refs.x;
                } else if (canS && (t.best.closest > nv.ES)) {
                    next = refs.y;
                } else {
                    next = lane_pop(t, stack);
                }
                if (next == RT_NONE) next = RT_TEAM_NEXT;
            } else {
                int remaining;
                const bool hit = lane_triangle_loaded(t, S.objects[t.obj].fwd, r0, r1, r2, tMin, remaining);
                if (hit && (meta & RT_TEAM_ANYHIT)) next = RT_NONE;
                else if (remaining <= 1) next = lane_pop(t, stack);
                else next = t.cur - 1;   // ~(k + 1)
                if (next == RT_NONE) next = RT_TEAM_NEXT;
            }
            t.cur = next;
        }
    }
#ifdef RT_SECTION_TIMERS
    if (lane == 0) {
        atomicAdd(&g_sectionCycles[0], dIter); atomicAdd(&g_sectionCycles[1], dLanes); atomicAdd(&g_sectionCycles[2], dIdle);
        atomicAdd(&g_sectionCycles[3], dJobs); atomicAdd(&g_sectionCycles[6], __builtin_amdgcn_s_memtime() - dStart);
    }
#endif
}

} // namespace

// Pathtrace (raytrace.comp:771-817) by workgroups of path waves and walker waves. Path waves take 8x8
// pixel tiles from a counter in device memory (zeroed by the launcher) until none is left.
__global__ __launch_bounds__(RT_TEAM_WAVES * 64) void RT_SYM(rt_teamkernel)(DevScene S, rt_push_constants pc, DevLaunch L, float4* __restrict__ accum,
                                                                           float4* __restrict__ out, unsigned long long* __restrict__ queue,
                                                                           int walkers, int refillMin)
{
    extern __shared__ uint2 stackMem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int pathWaves = RT_TEAM_WAVES - walkers;
    const TeamShared sh = team_shared(stackMem, pathWaves, walkers, S.stackDepth);
    for (int i = threadIdx.x; i < RT_TEAM_RING; i += RT_TEAM_WAVES * 64) sh.ring[i] = RT_TEAM_EMPTY;
    if (threadIdx.x < 16) { sh.ctrl[threadIdx.x] = 0; sh.pending[threadIdx.x] = 0; }
    __syncthreads();

    if (wave < walkers) {
        team_walker(S, sh, pathWaves, sh.stacks + (size_t)wave * S.stackDepth * 64 + lane, lane, refillMin);
        return;
    }

#ifdef RT_SECTION_TIMERS
    const unsigned long long teamStart = __builtin_amdgcn_s_memtime();
#endif
    TeamPath tp;
    tp.sh = sh;
    tp.wave = wave - walkers;
    tp.lane = lane;
    const int wavesX = L.tileSize >> 3;
    const int wavesPerTile = wavesX * wavesX;
    const unsigned long long totalWaves = (unsigned long long)L.ownedTiles * (unsigned long long)wavesPerTile;
    const int rpp = pc.raysPerPixel;
    const float raysPerPixelRatio = 1.0f / (float)rpp;
    const V3 sunDirN = normalize3(-ld3(pc.sunDirection));
    // parents may be tested concurrently when the scene allows it and one 64-bit mask can name every BVH parent
    const bool concurrent = S.orderFree != 0 && pc.parentBVHCount == S.objectCount && S.objectCount - S.anyLeafCount <= 64;
    for (;;) {
        unsigned long long item = 0;
        if (lane == 0) item = atomicAdd(queue, 1ull);
        item = (unsigned long long)__shfl((int)(unsigned)item, 0, 64) | ((unsigned long long)(unsigned)__shfl((int)(unsigned)(item >> 32), 0, 64) << 32);
        if (item >= totalWaves) break;
        const int ownedTile = (int)(item / (unsigned)wavesPerTile);
        const int sub = (int)(item - (unsigned long long)ownedTile * (unsigned)wavesPerTile);
        const int tile = L.tileRank + ownedTile * L.tileWorld;
        int tx, ty;
        rt_tile_of_position(tile, L.tilesX, &tx, &ty);
        const int sy = sub / wavesX, sx = sub - sy * wavesX;
        const int x = tx * L.tileSize + sx * 8 + (lane & 7);
        const int y = ty * L.tileSize + sy * 8 + (lane >> 3);
        const bool inside = x < L.width && y < L.height;
        const size_t idx = (size_t)(inside ? y : 0) * (size_t)L.width + (size_t)(inside ? x : 0);
        V3 nc = mk(0.0f, 0.0f, 0.0f);
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int fr = 0; fr < L.frames; ++fr) {
            const uint32_t frame = (uint32_t)(pc.frame + fr * L.frameStride);
            nc = mk(0.0f, 0.0f, 0.0f);
            for (int i = 0; i < rpp; ++i) {
                const uint32_t seed = seed_from_coords((uint32_t)x, (uint32_t)y, frame, (uint32_t)i);
                const Ray ray = get_ray(pc, x, y, seed);
                nc = nc + (concurrent ? ray_colour_team_async(S, pc, ray, inside, seed, sunDirN, tp)
                                      : ray_colour_team(S, pc, ray, inside, seed, sunDirN, tp));
            }
            if (inside) {
                acc = accum[idx];
                nc = nc * raysPerPixelRatio;
                acc.x += nc.x; acc.y += nc.y; acc.z += nc.z;
                accum[idx] = acc;
            }
        }
        if (inside) {
            const int lastFrame = pc.frame + (L.frames - 1) * L.frameStride;
            const float accumulationMultiplier = pc.accumulateFrames == 1 ? 1.0f / (float)(lastFrame + 1) : 1.0f;
            float4 avg = pc.accumulateFrames == 1 ? acc : make_float4(nc.x, nc.y, nc.z, 1.0f);
            avg.x *= accumulationMultiplier; avg.y *= accumulationMultiplier; avg.z *= accumulationMultiplier;
            out[idx] = make_float4(linear_to_gamma(avg.x), linear_to_gamma(avg.y), linear_to_gamma(avg.z), avg.w);
        }
    }
#ifdef RT_SECTION_TIMERS
    if (lane == 0) atomicAdd(&g_sectionCycles[5], __builtin_amdgcn_s_memtime() - teamStart);
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&sh.ctrl[2], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#if defined(RT_FLAVOUR_STRICT)
// LDS a workgroup of the team kernel needs: control words, ring, job slots of the path waves, the walkers' traversal stacks
extern "C" size_t rt_team_lds_bytes(int stackDepth, int walkers)
{
    const int pathWaves = RT_TEAM_WAVES - walkers;
    return 64 + RT_TEAM_RING * sizeof(int) + 64 + (size_t)pathWaves * RT_TEAM_SLOTS * (RT_TEAM_IN + RT_TEAM_OUT) * 64 * sizeof(float) +
           (size_t)walkers * (size_t)stackDepth * 64 * sizeof(uint2);
}
#else
extern "C" size_t rt_team_lds_bytes(int stackDepth, int walkers);
#endif

extern "C" hipError_t RT_SYM(rt_launch_teamkernel)(const DevScene* S, const rt_push_constants* pc, const DevLaunch* L, void* accum, void* out,
                                                   unsigned long long* queue, int workgroups, int walkers, int refillMin, hipStream_t stream)
{
    (void)hipGetLastError();      // an error left behind by an earlier, unrelated HIP call is not this launch's
    const int wavesX = L->tileSize / 8;
    const long long waves = (long long)L->ownedTiles * wavesX * wavesX;
    if (waves <= 0 || workgroups <= 0) return hipSuccess;
    if (walkers < 1 || walkers > RT_TEAM_WAVES - 1) return hipErrorInvalidValue;
    const int pathWaves = RT_TEAM_WAVES - walkers;
    const size_t lds = rt_team_lds_bytes(S->stackDepth, walkers);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(queue, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    // the attribute is raised once per size (per device: the largest size asked for so far)
    static thread_local size_t ldsSet[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > ldsSet[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(RT_SYM(rt_teamkernel)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        ldsSet[dev] = lds;
    }
    const long long need = (waves + pathWaves - 1) / pathWaves;
    const unsigned grid = (unsigned)(need < workgroups ? need : workgroups);
    hipLaunchKernelGGL(RT_SYM(rt_teamkernel), dim3(grid), dim3(RT_TEAM_WAVES * 64), lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out, queue,
                       walkers, refillMin < 1 ? 1 : (refillMin > 64 ? 64 : refillMin));
    return hipGetLastError();
}
