// Cross-lane building blocks for one 64-lane wavefront of gfx950: lane exchanges on the DPP /
// permlane-swap paths of the vector ALU (no LDS round trip), a 64-key sorting network built from
// them, and the strictly sequential fp32 sum across lanes that the in-search sigma-G clip needs.
#ifndef KB_WAVE_OPS_H_
#define KB_WAVE_OPS_H_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace kb {

// DPP controls of the GFX9 family (dst lane i reads src lane f(i) inside its row of 16 lanes;
// wave_* cross the rows).
constexpr int DPP_QUAD_XOR1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int DPP_QUAD_MIRROR = 0x1B;       // quad_perm:[3,2,1,0]
constexpr int DPP_ROW_ROR4 = 0x124;         // lane i <- lane (i - 4) mod 16
constexpr int DPP_ROW_ROR8 = 0x128;         // lane i <- lane (i - 8) mod 16 = i ^ 8
constexpr int DPP_ROW_ROR12 = 0x12C;        // lane i <- lane (i + 4) mod 16
constexpr int DPP_WAVE_SHL1 = 0x130;        // lane i <- lane i + 1
constexpr int DPP_WAVE_SHR1 = 0x138;        // lane i <- lane i - 1
constexpr int DPP_ROW_MIRROR = 0x140;       // lane i <- lane 15 - i
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane i <- lane 7 - i (inside its group of 8)

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_move(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// Value of lane (i ^ J).
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) {
    if constexpr (J == 1) {
        return dpp_move<DPP_QUAD_XOR1>(v);
    } else if constexpr (J == 2) {
        return dpp_move<DPP_QUAD_XOR2>(v);
    } else if constexpr (J == 4) {
        // banks {0, 2} of a row (lanes with bit 2 clear) read lane i + 4, banks {1, 3} lane i - 4
        int t = __builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_ROR12, 0xf, 0x5, false);
        t = __builtin_amdgcn_update_dpp(t, (int)v, DPP_ROW_ROR4, 0xf, 0xa, false);
        return (uint32_t)t;
    } else if constexpr (J == 8) {
        return dpp_move<DPP_ROW_ROR8>(v);
    } else if constexpr (J == 16) {
        // v_permlane16_swap: odd rows of the first operand <-> even rows of the second
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        static_assert(J == 32, "lane_xor distance");
        // v_permlane32_swap: upper half of the first operand <-> lower half of the second
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}

// Value of lane (i ^ (B - 1)): the lane mirrored inside its block of B.
template <int B>
__device__ __forceinline__ uint32_t lane_mirror(uint32_t v, int lane) {
    if constexpr (B == 2) {
        return dpp_move<DPP_QUAD_XOR1>(v);
    } else if constexpr (B == 4) {
        return dpp_move<DPP_QUAD_MIRROR>(v);
    } else if constexpr (B == 8) {
        return dpp_move<DPP_ROW_HALF_MIRROR>(v);
    } else if constexpr (B == 16) {
        return dpp_move<DPP_ROW_MIRROR>(v);
    } else if constexpr (B == 32) {
        return lane_xor<16>(dpp_move<DPP_ROW_MIRROR>(v), lane);
    } else {
        static_assert(B == 64, "lane_mirror block");
        return lane_xor<32>(lane_xor<16>(dpp_move<DPP_ROW_MIRROR>(v), lane), lane);
    }
}

// One compare-exchange step of an ascending network: the lane with BIT clear keeps the smaller key
// of (own, partner), the lane with BIT set the larger.  Strict comparisons on both sides: equal keys
// stay where they are, so the step is consistent without a tie-break (the payload travels along).
template <int BIT>
__device__ __forceinline__ void compare_exchange(uint32_t& key, uint32_t& pay, uint32_t pkey, uint32_t ppay, int lane) {
    const bool upper = (lane & BIT) != 0;
    const uint32_t lo = min(key, pkey), hi = max(key, pkey);
    const uint32_t nk = upper ? hi : lo;
    pay = (nk != key) ? ppay : pay;
    key = nk;
}

template <int B>
__device__ __forceinline__ void merge_block(uint32_t& key, uint32_t& pay, int lane) {
    // two sorted halves of a block of B -> sorted block: mirror step, then half-cleaners
    compare_exchange<B / 2>(key, pay, lane_mirror<B>(key, lane), lane_mirror<B>(pay, lane), lane);
    if constexpr (B >= 64) compare_exchange<16>(key, pay, lane_xor<16>(key, lane), lane_xor<16>(pay, lane), lane);
    if constexpr (B >= 32) compare_exchange<8>(key, pay, lane_xor<8>(key, lane), lane_xor<8>(pay, lane), lane);
    if constexpr (B >= 16) compare_exchange<4>(key, pay, lane_xor<4>(key, lane), lane_xor<4>(pay, lane), lane);
    if constexpr (B >= 8) compare_exchange<2>(key, pay, lane_xor<2>(key, lane), lane_xor<2>(pay, lane), lane);
    if constexpr (B >= 4) compare_exchange<1>(key, pay, lane_xor<1>(key, lane), lane_xor<1>(pay, lane), lane);
}

// Ascending sort of the 64 (key, payload) pairs held one per lane; 21 compare-exchange steps,
// 18 of them single- or double-DPP moves, 3 through the permlane swaps.  Equal keys end up in an
// unspecified relative order.  All 64 lanes must be active.
__device__ __forceinline__ void wave_sort64(uint32_t& key, uint32_t& pay, int lane) {
    merge_block<2>(key, pay, lane);
    merge_block<4>(key, pay, lane);
    merge_block<8>(key, pay, lane);
    merge_block<16>(key, pay, lane);
    merge_block<32>(key, pay, lane);
    merge_block<64>(key, pay, lane);
}

// Sorting more than 64 keys with one wavefront: E keys per lane, key position p = slot * 64 + lane.
// The classic bitonic network on 64 * E positions: distances below 64 exchange between lanes (the moves
// above), distances of 64 and more between the registers of one lane.
template <int J>
__device__ __forceinline__ void exchange_between_lanes(uint32_t& key, uint32_t& pay, bool descending, int lane) {
    const uint32_t pkey = lane_xor<J>(key, lane), ppay = lane_xor<J>(pay, lane);
    const bool keep_larger = ((lane & J) != 0) != descending;
    const uint32_t lo = min(key, pkey), hi = max(key, pkey);
    const uint32_t nk = keep_larger ? hi : lo;
    pay = (nk != key) ? ppay : pay;
    key = nk;
}

template <int E>
__device__ __forceinline__ void wave_sort_multi(uint32_t (&key)[E], uint32_t (&pay)[E], int lane) {
    static_assert(E == 2 || E == 4, "keys per lane");
    constexpr int N = 64 * E;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {  // partners are registers of the same lane: slots s and s ^ (j / 64)
                constexpr int unused = 0;
                (void)unused;
#pragma unroll
                for (int s = 0; s < E; ++s) {
                    const int t = s ^ (j >> 6);
                    if (t > s) {
                        const bool descending = ((s * 64) & k) != 0;  // (p & k), the lane bits are below 64 <= j < k
                        const bool swap = descending ? (key[s] < key[t]) : (key[s] > key[t]);
                        const uint32_t ks = swap ? key[t] : key[s], kt = swap ? key[s] : key[t];
                        const uint32_t ps = swap ? pay[t] : pay[s], pt = swap ? pay[s] : pay[t];
                        key[s] = ks;
                        key[t] = kt;
                        pay[s] = ps;
                        pay[t] = pt;
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < E; ++s) {
                    const bool descending = (((s * 64) | lane) & k) != 0;
                    if (j == 32) exchange_between_lanes<32>(key[s], pay[s], descending, lane);
                    if (j == 16) exchange_between_lanes<16>(key[s], pay[s], descending, lane);
                    if (j == 8) exchange_between_lanes<8>(key[s], pay[s], descending, lane);
                    if (j == 4) exchange_between_lanes<4>(key[s], pay[s], descending, lane);
                    if (j == 2) exchange_between_lanes<2>(key[s], pay[s], descending, lane);
                    if (j == 1) exchange_between_lanes<1>(key[s], pay[s], descending, lane);
                }
            }
        }
    }
}

__device__ __forceinline__ uint32_t lane_next(uint32_t v) {  // lane i <- lane i + 1 (lane 63: 0)
    return dpp_move<DPP_WAVE_SHL1>(v);
}

// acc[i] <- acc[i - 1] + y[i] on every lane.  Repeated m times from acc = 0, lane i ends up with
// ((0 + y[i-m+1]) + y[i-m+2]) + ... + y[i], one correctly rounded fp32 add after the other: the
// sequential sum of m consecutive lanes without moving a value through a scalar register.
__device__ __forceinline__ float chain_add(float acc, float y) {
    const float prev = __uint_as_float(dpp_move<DPP_WAVE_SHR1>(__float_as_uint(acc)));
    return prev + y;
}
// The same step with a carry entering at lane 0 (the running sum of the 64 positions before this slot).
// minimum / maximum of one int per lane over the wave, in a scalar register
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ float chain_add_carry(float acc, float y, float carry) {
    const float prev = __int_as_float(
            __builtin_amdgcn_update_dpp(__float_as_int(carry), __float_as_int(acc), DPP_WAVE_SHR1, 0xf, 0xf, false));
    return prev + y;
}

}  // namespace kb
#endif
