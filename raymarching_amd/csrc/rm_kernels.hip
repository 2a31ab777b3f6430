// rm_kernels.hip -- gfx950 kernels of librmhip: the fused per-pixel march
// (primary ray + central-difference normal + colour + soft-shadow march + shade)
// that replaces pass 0 of the reference's fragment shader
// (Assets/Raymarching.shader:149-216 with Assets/Raymarching.cginc:334-392), and the
// depth-only passes built from raymarch().
#include <hip/hip_runtime.h>
#include "rm_device.h"
#include "rm_math.h"
#include "rm_launch.h"

// block geometry of the tile kernel: RM_BWX x RM_BWY waves, each an 8x8 pixel tile
#ifndef RM_BWX
#define RM_BWX 2
#endif
#ifndef RM_BWY
#define RM_BWY 2
#endif
#define RM_BLOCK (64 * RM_BWX * RM_BWY)
// words per block in the counter-slot buffer; a timeline build (developer tool) appends
// start/end timestamps of the block (100 MHz s_memrealtime) and its hardware id
#ifdef RM_TIMELINE
#define RM_SLOT_WORDS 8
#else
#define RM_SLOT_WORDS 3
#endif
#ifdef RM_MIN_WAVES
#define RM_LB __launch_bounds__(RM_BLOCK, RM_MIN_WAVES)
#else
#define RM_LB __launch_bounds__(RM_BLOCK)
#endif

namespace {

struct MarchResult { bool hit; int evals; float length; float contact; };

// cginc:367-392.  `steps` = min(2000,_MaxSteps), `depth` already min'ed with _MaxDist.
template <class S>
RM_DI MarchResult rm_raymarch(const S &sc, V3 ro, V3 rd, float depth, float steps, float sf) {
    const float kc = rm_vgpr(.0025f);
    MarchResult r;
    r.hit = false; r.evals = 0; r.contact = 0.0f;
    float dm = 0.0f;
    for (int i = 0; (float)i < steps; i++) {
        V3 cp = mk3(ro.x + rd.x * dm, ro.y + rd.y * dm, ro.z + rd.z * dm);
        float dts = sc.w(cp);
        r.evals++;
        dm += fabsf(dts) * sf;
        r.contact = dm * kc;
        if (fabsf(dts) < r.contact) { r.hit = true; break; }
        if (dm > depth) break;
    }
    r.length = dm;
    return r;
}

// shader:102-123 with k = 16
template <class S>
RM_DI float rm_getshadow(const S &sc, V3 p, V3 n, V3 L, float contact, float depth, float steps, float sf, int &evals) {
    const float kc = rm_vgpr(.0025f);
    float dm = 0.0f, res = 1.0f;
    p = mk3(p.x + (n.x * contact) * 2.0f, p.y + (n.y * contact) * 2.0f, p.z + (n.z * contact) * 2.0f);
    int ev = 0;
    for (int i = 0; (float)i < steps; i++) {
        V3 cp = mk3(p.x + L.x * dm, p.y + L.y * dm, p.z + L.z * dm);
        float dts = sc.w(cp);
        ev++;
        res = fminf(res, 16.0f * dts / dm);
        dm += fabsf(dts) * sf;
        contact = dm * kc;
        if (fabsf(dts) < contact) break;
        if (dm > depth) break;
    }
    evals = ev;
    return res;
}

// One thread = one pixel.  A wave covers an 8x8 pixel tile (neighbouring rays take
// similar step counts), a 256-thread block a 16x16 tile.  blockIdx.z = pose.
template <class S>
RM_DI void rm_pixel(const RmParams &U, const S &sc) {
    __shared__ unsigned int s_cnt[3];
    const int tid = threadIdx.x;
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();
    unsigned c_hit = 0, c_p = 0, c_s = 0;
#ifdef RM_TIMELINE
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif

    const int lane = tid & 63, wv = tid >> 6;
    const int x  = blockIdx.x * (8 * RM_BWX) + (wv % RM_BWX) * 8 + (lane & 7);
#ifdef RM_EXPERIMENT_ROWSHIFT
    const int by_ = (int)((blockIdx.y + (unsigned)RM_EXPERIMENT_ROWSHIFT) % gridDim.y);
#else
    const int by_ = (int)blockIdx.y;
#endif
    const int yl = by_ * (8 * RM_BWY) + (wv / RM_BWX) * 8 + (lane >> 3);
    const int pose = blockIdx.z;
    const bool valid = x < U.w && yl < U.rows;

    if (valid) {
        // compact local row -> frame row (row-band shard; identity when n_parts == 1)
        const int band = yl / U.band_rows;
        const int y = (band * U.n_parts + U.part) * U.band_rows + (yl - band * U.band_rows);
        const size_t gidx = (size_t)y * (size_t)U.w + (size_t)x;                    // full-frame index
        const size_t oidx = ((size_t)pose * (size_t)U.rows + (size_t)yl) * (size_t)U.w + (size_t)x;

        const RmView &V = U.views ? U.views[pose] : U.view;
        const float uu = ((float)x + 0.5f) / (float)U.w;
        const float vv = ((float)y + 0.5f) / (float)U.h;
        V3 ro, rd;
        if (U.mode == RM_MODE_DEPTH_ORTHO) {
            V3 c0 = mk3(U.ortho_c[0][0], U.ortho_c[0][1], U.ortho_c[0][2]);
            V3 c1 = mk3(U.ortho_c[1][0], U.ortho_c[1][1], U.ortho_c[1][2]);
            V3 c2 = mk3(U.ortho_c[2][0], U.ortho_c[2][1], U.ortho_c[2][2]);
            V3 c3 = mk3(U.ortho_c[3][0], U.ortho_c[3][1], U.ortho_c[3][2]);
            ro = rm_lerp3(rm_lerp3(c3, c2, uu), rm_lerp3(c0, c1, uu), vv);
            rd = mk3(U.ortho_fwd[0], U.ortho_fwd[1], U.ortho_fwd[2]);
        } else {
            // rasteriser interpolation of vert's viewDir (cginc:26-33) to the pixel centre
            V3 bl = mk3(V.bl[0], V.bl[1], V.bl[2]), br = mk3(V.br[0], V.br[1], V.br[2]);
            V3 tl = mk3(V.tl[0], V.tl[1], V.tl[2]), tr = mk3(V.tr[0], V.tr[1], V.tr[2]);
            ro = mk3(rm_vgpr(V.ro[0]), rm_vgpr(V.ro[1]), rm_vgpr(V.ro[2]));
            rd = rm_lerp3(rm_lerp3(bl, br, uu), rm_lerp3(tl, tr, uu), vv);
        }

        float depth = U.far_plane;                                    // shader:154
        if (U.mode == RM_MODE_FRAG && U.sdepth) depth = U.sdepth[gidx];
        depth = fminf(depth, U.max_dist);                             // cginc:368

        const float sf = rm_vgpr(U.step_factor);
        MarchResult r = rm_raymarch(sc, ro, rd, depth, U.steps, sf);
        c_p = (unsigned)r.evals;
        c_hit = r.hit ? 1u : 0u;

        if (U.mode == RM_MODE_FRAG) {
            uint32_t rgba;
            if (r.hit) {
                const V3 L = mk3(rm_vgpr(U.light[0]), rm_vgpr(U.light[1]), rm_vgpr(U.light[2]));
                V3 hp = mk3(ro.x + rd.x * r.length, ro.y + rd.y * r.length, ro.z + rd.z * r.length);   // shader:174
                const float e = U.nss;
                V3 raw = mk3(sc.w(mk3(hp.x + e, hp.y, hp.z)) - sc.w(mk3(hp.x - e, hp.y, hp.z)),        // shader:63-67
                             sc.w(mk3(hp.x, hp.y + e, hp.z)) - sc.w(mk3(hp.x, hp.y - e, hp.z)),
                             sc.w(mk3(hp.x, hp.y, hp.z + e)) - sc.w(mk3(hp.x, hp.y, hp.z - e)));
                V3 n = rm_normalize(raw);
                V4 s = sc.full(hp);                                                                   // shader:193
                int sev = 0;
                float sh = rm_getshadow(sc, hp, n, L, r.contact, U.shadow_depth, U.steps, sf, sev);
                c_s = (unsigned)sev;
                float shadow = sh * 0.75f + .25f;                        // remap(x,0,1,.25,1), cginc:80-82
                float ndl = rm_dot3(n, L) * .5f + .5f;
                float ao = rm_clamp01(rm_len3(mk3(raw.x / U.nss_raw, raw.y / U.nss_raw, raw.z / U.nss_raw)));  // shader:125-127
                float cr = ((s.x * ndl) * shadow) * ao;
                float cg = ((s.y * ndl) * shadow) * ao;
                float cb = ((s.z * ndl) * shadow) * ao;
                rgba = rm_unorm8(cr) | (rm_unorm8(cg) << 8) | (rm_unorm8(cb) << 16) | 0xFF000000u;
            } else if (U.src) {
                rgba = reinterpret_cast<const uint32_t *>(U.src)[gidx] | 0xFF000000u;
            } else {
                rgba = U.bg_rgba8;
            }
            if (U.dst_rgba) reinterpret_cast<uint32_t *>(U.dst_rgba)[oidx] = rgba;
        }
        if (U.dst_depth) U.dst_depth[oidx] = r.length;
        if (U.dst_hit) U.dst_hit[oidx] = r.hit ? 1 : 0;
    }

    // Counters: per-wave sums by cross-lane shuffles, per-block sums in LDS, then ONE plain
    // store per counter into this block's own slot.  No global atomics: same-address atomics
    // from thousands of short blocks serialise in L2 (~6 ns each) and would bound the frame.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c_hit += __shfl_xor(c_hit, o);
        c_p += __shfl_xor(c_p, o);
        c_s += __shfl_xor(c_s, o);
    }
    if (lane == 0) {
        if (c_hit) atomicAdd(&s_cnt[0], c_hit);
        if (c_p) atomicAdd(&s_cnt[1], c_p);
        if (c_s) atomicAdd(&s_cnt[2], c_s);
    }
    __syncthreads();
    if (tid < 3) {
        const unsigned bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;   // slot by launch index, not by image position
        U.partials[RM_SLOT_WORDS * bid + tid] = s_cnt[tid];
#ifdef RM_TIMELINE
        if (tid == 0) {
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned *o = U.partials + RM_SLOT_WORDS * bid;
            o[3] = (unsigned)t_start; o[4] = (unsigned)(t_start >> 32); o[5] = (unsigned)t_end; o[6] = (unsigned)(t_end >> 32);
            o[7] = (hwid & 0xffffu) | (xcc << 16);
        }
#endif
    }
}

// sums the per-block counter slots of the last launch: out[0..2] = n_hit, primary evals, shadow evals
__global__ __launch_bounds__(1024) void rm_reduce_stats_kernel(const unsigned *__restrict__ partials, unsigned n_blocks,
                                                               unsigned long long *__restrict__ out) {
    __shared__ unsigned long long s[3];
    if (threadIdx.x < 3) s[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long a0 = 0, a1 = 0, a2 = 0;
    for (unsigned b = threadIdx.x; b < n_blocks; b += 1024) {
        a0 += partials[RM_SLOT_WORDS * b]; a1 += partials[RM_SLOT_WORDS * b + 1]; a2 += partials[RM_SLOT_WORDS * b + 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s[0], a0); atomicAdd(&s[1], a1); atomicAdd(&s[2], a2); }
    __syncthreads();
    if (threadIdx.x < 3) out[threadIdx.x] = s[threadIdx.x];
}

template <bool ROT, int SCL>
__global__ RM_LB void rm_tile_kernel4(const RmParams U) {
    Scene4<ROT, SCL> sc(U);
    rm_pixel(U, sc);
}

template <bool ROT, int SCL>
__global__ RM_LB void rm_tile_kernelN(const RmParams U) {
    SceneN<ROT, SCL> sc(U.prim_table, U.n_prims);
    rm_pixel(U, sc);
}

template <bool ROT, int SCL>
hipError_t launch_variant(const RmParams &U, dim3 grid, hipStream_t stream) {
    if (U.n_prims > 0) hipLaunchKernelGGL((rm_tile_kernelN<ROT, SCL>), grid, dim3(RM_BLOCK), 0, stream, U);
    else               hipLaunchKernelGGL((rm_tile_kernel4<ROT, SCL>), grid, dim3(RM_BLOCK), 0, stream, U);
    return hipGetLastError();
}

}  // namespace

unsigned rm_slot_words() { return RM_SLOT_WORDS; }

unsigned rm_tile_blocks(int w, int rows, int n_poses) {
    return (unsigned)((w + 8 * RM_BWX - 1) / (8 * RM_BWX)) * (unsigned)((rows + 8 * RM_BWY - 1) / (8 * RM_BWY)) * (unsigned)n_poses;
}

hipError_t rm_reduce_stats(const unsigned *partials, unsigned n_blocks, unsigned long long *out, hipStream_t stream) {
    hipLaunchKernelGGL(rm_reduce_stats_kernel, dim3(1), dim3(1024), 0, stream, partials, n_blocks, out);
    return hipGetLastError();
}

hipError_t rm_launch_tiles(const RmParams &U, hipStream_t stream) {
    if (U.w <= 0 || U.rows <= 0 || U.n_poses <= 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)((U.w + 8 * RM_BWX - 1) / (8 * RM_BWX)), (unsigned)((U.rows + 8 * RM_BWY - 1) / (8 * RM_BWY)), (unsigned)U.n_poses);
    if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidValue;
    const bool rot = !(U.scene_flags & RM_SF_IDENT_ROT);
    const int scl = (U.scene_flags & RM_SF_UNIT_SCALE) ? RM_SCL_UNIT : (U.scene_flags & RM_SF_RCP) ? RM_SCL_RCP : RM_SCL_DIV;
    if (rot) {
        if (scl == RM_SCL_UNIT) return launch_variant<true, RM_SCL_UNIT>(U, grid, stream);
        if (scl == RM_SCL_RCP)  return launch_variant<true, RM_SCL_RCP>(U, grid, stream);
        return launch_variant<true, RM_SCL_DIV>(U, grid, stream);
    }
    if (scl == RM_SCL_UNIT) return launch_variant<false, RM_SCL_UNIT>(U, grid, stream);
    if (scl == RM_SCL_RCP)  return launch_variant<false, RM_SCL_RCP>(U, grid, stream);
    return launch_variant<false, RM_SCL_DIV>(U, grid, stream);
}
