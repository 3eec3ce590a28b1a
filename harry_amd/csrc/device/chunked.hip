// HIP kernels of the chunked profile (.hry v0.2): every (context plane, chunk) is an independent stream with a
// fresh adaptive model and a fresh 64-bit Moffat-Neal-Witten coder -- the reference's arithmetic
// (arith/coder.h:58-162, arith/stat_adaptive.h:46-90) instantiated once per chunk.
//
//   k_chunk_encode : one wavefront owns one stream: count / cumulative tables in LDS, 64 symbols per step evaluated
//                    by counting, range recurrence in scalar registers, low register accumulated through an LDS
//                    window into per-stream big-number accumulators
//   k_stream_offsets + k_pack_streams : wavefront prefix scan of the stream byte lengths, coalesced packing
//   k_chunk_decode : one wavefront owns one stream: inclusive-cumulative table in registers (4 entries per lane),
//                    symbol search by wave-wide compare + ballot (no division by the data-dependent r)
#include <hip/hip_runtime.h>

#include "codec_math.hpp"
#include "dev_types.hpp"
#include "kernels.hpp"

namespace hry {
namespace dev {

__device__ __forceinline__ uint32_t wscan_excl(uint32_t v, uint32_t &total)
{
	uint32_t inc = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t o = __shfl_up(inc, d, 64);
		if ((int)(threadIdx.x & 63) >= d) inc += o;
	}
	total = __shfl(inc, 63, 64);
	return inc - v;
}

constexpr int kWin = 192;   // LDS accumulation window in 32-bit words (64 symbols x <= 63 shifts = 126 words + 3)

__global__ __launch_bounds__(64) void k_chunk_encode(const StreamJob *jobs, const uint32_t *inits, const MagicEnt *magic,
                                                     unsigned long long *acc, uint32_t *stream_bits)
{
	const StreamJob jb = jobs[blockIdx.x];
	const int lane = threadIdx.x;
	__shared__ uint32_t cnt[256], cum[256], bh[256];
	__shared__ unsigned long long win[kWin];
	{
		const uint32_t *st = inits + (size_t)jb.init * 256 + 4 * lane;
		uint32_t a = st[0], b = st[1], c = st[2], d = st[3], tot;
		uint32_t ex = wscan_excl(a + b + c + d, tot);
		cnt[4 * lane] = a; cnt[4 * lane + 1] = b; cnt[4 * lane + 2] = c; cnt[4 * lane + 3] = d;
		cum[4 * lane] = ex; cum[4 * lane + 1] = ex + a; cum[4 * lane + 2] = ex + a + b; cum[4 * lane + 3] = ex + a + b + c;
	}
	__syncthreads();
	uint64_t R = 1ull << 63;   // coder.h:47
	uint32_t S = 0;
	for (uint32_t base = 0; base < jb.n; base += 64) {
		const uint32_t j = base + lane;
		const bool valid = j < jb.n;
		const uint32_t nb = min(64u, jb.n - base);
		uint32_t s = valid ? jb.sym[j] : 0x100u;
		uint32_t l = valid ? cum[s] : 0, c = valid ? cnt[s] : 0;
		for (uint32_t i = 0; i < nb; ++i) {
			uint32_t si = (uint32_t)__builtin_amdgcn_readlane(s, i);
			if ((int)i < lane) { l += si < s ? 1u : 0u; c += si == s ? 1u : 0u; }
		}
		// per-symbol constants of the recurrence
		uint32_t t = jb.t0 + j;
		bool sub = valid && (l + c == t);
		bool noop = !valid || (sub && l == 0);
		MagicEnt me = valid ? magic[t] : MagicEnt{ 0, 0, 0 };
		uint32_t mlo = (uint32_t)me.magic, mhi = (uint32_t)(me.magic >> 32), mx = sub ? l : c;
		uint32_t mm = me.shift | (sub ? kMetaSub : 0u) | (noop ? kMetaNoop : 0u);
		// serial recurrence over the batch (arith/coder.h:69-91 without the low register)
		uint64_t my_r = 0;
		uint32_t my_s = 0;
		const uint32_t s_first = S;
		for (uint32_t i = 0; i < nb; ++i) {
			uint32_t meta = (uint32_t)__builtin_amdgcn_readlane(mm, i);
			uint64_t r = 0;
			uint32_t s_before = S;
			if (!(meta & kMetaNoop)) {
				uint64_t mg = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(mhi, i) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(mlo, i);
				uint32_t x = (uint32_t)__builtin_amdgcn_readlane(mx, i);
				r = cm::div_by_magic(R, mg, meta & 63u);
				uint64_t prod = r * x;
				uint64_t Rn = (meta & kMetaSub) ? R - prod : prod;
				uint64_t y = Rn - 1;
				uint32_t sh = (y ? (uint32_t)__builtin_clzll(y) : 64u) - 1u;
				R = Rn << sh;
				S += sh;
			}
			if (lane == (int)i) { my_r = r; my_s = s_before; }
		}
		// low register: L += r * l at bit position my_s (coder.h:71), gathered in an LDS window first
		const uint32_t w0 = s_first >> 5, span = ((S + 95) >> 5) - w0 + 1;
		uint64_t a = (valid && l) ? my_r * l : 0;
		uint32_t w = my_s >> 5, shb = my_s & 31;
		uint64_t hi = a >> (32 + shb), low = a << (32 - shb);
		uint32_t mid = (uint32_t)(low >> 32), lo = (uint32_t)low;
		unsigned long long *dst = acc + jb.word_base;
		if (span <= (uint32_t)kWin) {
			for (uint32_t k = lane; k < span; k += 64) win[k] = 0;
			bh[4 * lane] = 0; bh[4 * lane + 1] = 0; bh[4 * lane + 2] = 0; bh[4 * lane + 3] = 0;
			__syncthreads();
			if (a) {
				if (hi) atomicAdd(&win[w - w0], (unsigned long long)hi);
				if (mid) atomicAdd(&win[w - w0 + 1], (unsigned long long)mid);
				if (lo) atomicAdd(&win[w - w0 + 2], (unsigned long long)lo);
			}
			if (valid) atomicAdd(&bh[s], 1u);
			__syncthreads();
			for (uint32_t k = lane; k < span; k += 64) {
				unsigned long long v = win[k];
				if (v) atomicAdd(&dst[w0 + k], v);
			}
		} else {
			bh[4 * lane] = 0; bh[4 * lane + 1] = 0; bh[4 * lane + 2] = 0; bh[4 * lane + 3] = 0;
			__syncthreads();
			if (a) {
				if (hi) atomicAdd(&dst[w], (unsigned long long)hi);
				if (mid) atomicAdd(&dst[w + 1], (unsigned long long)mid);
				if (lo) atomicAdd(&dst[w + 2], (unsigned long long)lo);
			}
			if (valid) atomicAdd(&bh[s], 1u);
			__syncthreads();
		}
		// adaptive update of the tables by the whole batch (stat_adaptive.h:77-82)
		uint32_t a0 = bh[4 * lane], a1 = bh[4 * lane + 1], a2 = bh[4 * lane + 2], a3 = bh[4 * lane + 3], tot;
		uint32_t ex = wscan_excl(a0 + a1 + a2 + a3, tot);
		cnt[4 * lane] += a0; cnt[4 * lane + 1] += a1; cnt[4 * lane + 2] += a2; cnt[4 * lane + 3] += a3;
		cum[4 * lane] += ex; cum[4 * lane + 1] += ex + a0; cum[4 * lane + 2] += ex + a0 + a1; cum[4 * lane + 3] += ex + a0 + a1 + a2;
		__syncthreads();
	}
	if (lane == 0) stream_bits[blockIdx.x] = S + 64;   // flush: the 64 bits of the low register (coder.h:58-67)
}

// byte length of every stream and exclusive prefix (one workgroup; wave scans + LDS for the wave totals)
__global__ __launch_bounds__(1024) void k_stream_offsets(const uint32_t *stream_bits, uint32_t n, uint32_t *nbytes, unsigned long long *offsets)
{
	__shared__ unsigned long long wave_tot[16];
	__shared__ unsigned long long carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (uint32_t base = 0; base < n; base += 1024) {
		uint32_t i = base + threadIdx.x;
		uint32_t nb = i < n ? (stream_bits[i] + 7) >> 3 : 0;
		unsigned long long inc = nb;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			unsigned long long o = __shfl_up(inc, d, 64);
			if (lane >= d) inc += o;
		}
		if (lane == 63) wave_tot[wv] = inc;
		__syncthreads();
		unsigned long long pre = carry;
		for (int k = 0; k < wv; ++k) pre += wave_tot[k];
		if (i < n) { nbytes[i] = nb; offsets[i] = pre + inc - nb; }
		__syncthreads();
		if (threadIdx.x == 1023) carry = pre + inc;
		__syncthreads();
	}
	if (threadIdx.x == 0) offsets[n] = carry;
}

__global__ __launch_bounds__(256) void k_pack_streams(const StreamJob *jobs, const uint8_t *bytes, const uint32_t *nbytes, const unsigned long long *offsets, uint8_t *out)
{
	const StreamJob jb = jobs[blockIdx.x];
	const uint8_t *src = bytes + (size_t)jb.word_base * 4;
	uint8_t *dst = out + offsets[blockIdx.x];
	for (uint32_t i = threadIdx.x; i < nbytes[blockIdx.x]; i += 256) dst[i] = src[i];
}

// op symbols -> one plane per order class (host sends symbol + class per operation and the position inside its class)
__global__ __launch_bounds__(256) void k_scatter_u8(const uint8_t *src, const uint32_t *dst_index, uint32_t n, uint8_t *dst)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[dst_index[i]] = src[i];
}

// ---------------------------------------------------------------------------------------------------------
// decode: coder.h:124-162 + stat_adaptive.h:55-72.  target = min(t - 1, D / r) and the Fenwick descent are
// replaced by their definition: the symbol is the number of table entries whose inclusive cumulative count I
// satisfies I <= target  <=>  I * r <= D and I < t.  Each lane holds 4 consecutive entries.
// ---------------------------------------------------------------------------------------------------------
struct BitFeed {
	const uint8_t *p;
	uint32_t nbytes, pos;   // pos = next byte
	__device__ __forceinline__ uint64_t take64()   // next 8 bytes, big-endian, 0xFF past the end (bitstream.h:27)
	{
		uint64_t v = 0;
		for (int k = 0; k < 8; ++k) { uint32_t b = pos < nbytes ? p[pos] : 0xffu; ++pos; v = (v << 8) | b; }
		return v;
	}
};

__global__ __launch_bounds__(64) void k_chunk_decode(const StreamJob *jobs, const uint32_t *inits, const MagicEnt *magic,
                                                     const uint8_t *payload, const unsigned long long *offsets, const uint32_t *nbytes, uint8_t *sym_out_base)
{
	const StreamJob jb = jobs[blockIdx.x];
	const int lane = threadIdx.x;
	uint8_t *out = const_cast<uint8_t*>(jb.sym);
	(void)sym_out_base;
	// inclusive cumulative counts and counts of entries 4*lane .. 4*lane+3
	uint32_t c0, c1, c2, c3, i0, i1, i2, i3;
	{
		const uint32_t *st = inits + (size_t)jb.init * 256 + 4 * lane;
		c0 = st[0]; c1 = st[1]; c2 = st[2]; c3 = st[3];
		uint32_t tot, ex = wscan_excl(c0 + c1 + c2 + c3, tot);
		i0 = ex + c0; i1 = i0 + c1; i2 = i1 + c2; i3 = i2 + c3;
	}
	BitFeed bf{ payload + offsets[blockIdx.x], nbytes[blockIdx.x], 0 };
	uint64_t D = bf.take64();          // coder.h:124-129
	uint64_t buf = bf.take64();        // look-ahead bits, consumed from the top
	uint32_t buf_bits = 64;
	uint64_t R = 1ull << 63;
	for (uint32_t base = 0; base < jb.n; base += 64) {
		const uint32_t nb = min(64u, jb.n - base);
		// reciprocals of the totals of the next 64 symbols: t is known in advance (t0 + position)
		MagicEnt me = base + lane < jb.n ? magic[jb.t0 + base + lane] : MagicEnt{ 0, 0, 0 };
		const uint32_t mlo = (uint32_t)me.magic, mhi = (uint32_t)(me.magic >> 32), msh = me.shift;
		uint32_t mysym = 0;
		for (uint32_t i = 0; i < nb; ++i) {
			const uint32_t t = jb.t0 + base + i;
			uint64_t mg = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(mhi, i) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(mlo, i);
			uint32_t shf = (uint32_t)__builtin_amdgcn_readlane(msh, i);
			uint64_t r = t >= 2 ? cm::div_by_magic(R, mg, shf) : R;
			// symbol = #{ entries : I * r <= D and I < t }; I * r cannot overflow: I < t, r = floor(R / t) => I * r < R <= 2^63
			bool b0 = i0 < t && (uint64_t)i0 * r <= D, b1 = i1 < t && (uint64_t)i1 * r <= D, b2 = i2 < t && (uint64_t)i2 * r <= D, b3 = i3 < t && (uint64_t)i3 * r <= D;
			uint32_t below = (b0 ? 1u : 0u) + (b1 ? 1u : 0u) + (b2 ? 1u : 0u) + (b3 ? 1u : 0u);
			uint64_t full = __ballot(below == 4);
			uint32_t lanes_full = (uint32_t)__popcll(full);          // entries are non-decreasing: full lanes form a prefix
			uint32_t part = (uint32_t)__builtin_amdgcn_readlane(below, lanes_full < 64 ? lanes_full : 63);
			uint32_t s = lanes_full < 64 ? lanes_full * 4 + part : 255u;
			// l = inclusive count of entry s-1, h = inclusive count of entry s
			uint32_t src_lane = s >> 2, k = s & 3;
			uint32_t incl = k == 0 ? i0 : k == 1 ? i1 : k == 2 ? i2 : i3;
			uint32_t cn = k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : c3;
			uint32_t h = (uint32_t)__builtin_amdgcn_readlane(incl, src_lane);
			uint32_t l = h - (uint32_t)__builtin_amdgcn_readlane(cn, src_lane);
			// coder.h:140-153
			D -= r * l;
			uint64_t Rn = h < t ? r * (uint64_t)(h - l) : R - r * l;
			uint64_t y = Rn - 1;
			uint32_t sh = (y ? (uint32_t)__builtin_clzll(y) : 64u) - 1u;
			R = Rn << sh;
			uint32_t take = sh;
			while (take) {   // shift in the next bits of the stream
				if (buf_bits == 0) { buf = bf.take64(); buf_bits = 64; }
				uint32_t n = take < buf_bits ? take : buf_bits;
				D = n == 64 ? buf : (D << n) | (buf >> (64 - n));
				buf = n == 64 ? 0 : buf << n;
				buf_bits -= n;
				take -= n;
			}
			// adaptive update (stat_adaptive.h:77-82): count of s, inclusive counts of every entry >= s
			const uint32_t e = 4 * lane;
			c0 += (e == s); c1 += (e + 1 == s); c2 += (e + 2 == s); c3 += (e + 3 == s);
			i0 += (e >= s); i1 += (e + 1 >= s); i2 += (e + 2 >= s); i3 += (e + 3 >= s);
			if (lane == (int)i) mysym = s;
		}
		if (base + lane < jb.n) out[base + lane] = (uint8_t)mysym;
	}
}

// ---------------------------------------------------------------------------------------------------------
void launch_chunk_encode(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *inits, const MagicEnt *magic, uint64_t *acc, uint32_t *stream_bits)
{
	if (nstreams) hipLaunchKernelGGL(k_chunk_encode, dim3(nstreams), dim3(64), 0, st, jobs, inits, magic, (unsigned long long*)acc, stream_bits);
}
void launch_stream_pack(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *stream_bits, const uint8_t *bytes,
                        uint32_t *nbytes, uint64_t *offsets, uint8_t *out, bool pack)
{
	if (!pack) hipLaunchKernelGGL(k_stream_offsets, dim3(1), dim3(1024), 0, st, stream_bits, nstreams, nbytes, (unsigned long long*)offsets);
	else if (nstreams) hipLaunchKernelGGL(k_pack_streams, dim3(nstreams), dim3(256), 0, st, jobs, bytes, nbytes, (const unsigned long long*)offsets, out);
}
void launch_scatter_u8(hipStream_t st, const uint8_t *src, const uint32_t *dst_index, uint32_t n, uint8_t *dst)
{
	if (n) hipLaunchKernelGGL(k_scatter_u8, dim3((n + 255) / 256), dim3(256), 0, st, src, dst_index, n, dst);
}
void launch_chunk_decode(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *inits, const MagicEnt *magic,
                         const uint8_t *payload, const uint64_t *offsets, const uint32_t *nbytes)
{
	if (nstreams) hipLaunchKernelGGL(k_chunk_decode, dim3(nstreams), dim3(64), 0, st, jobs, inits, magic, payload, (const unsigned long long*)offsets, nbytes, (uint8_t*)nullptr);
}

}   // namespace dev
}   // namespace hry
