// HIP kernels of the chunked profile (.hry v0.2): every (context plane, chunk) is an independent stream with a
// fresh adaptive model and a fresh Moffat-Neal-Witten coder with 32-bit registers -- the reference's templates
// arith::Encoder<uint32_t> / arith::Decoder<uint32_t> (arith/coder.h:27-162) and arith/stat_adaptive.h:46-90 instantiated
// once per chunk.  32-bit registers are what the scalar unit multiplies in one instruction; a chunk holds at most 2^20
// symbols, so totals stay below 2^21 and the interval keeps >= 10 bits per count (size effect < 0.01 %).
//
//   k_chunk_encode : one wavefront owns one stream: count / cumulative tables in LDS, 64 symbols per step evaluated
//                    by counting, range recurrence in scalar registers, low register accumulated through an LDS
//                    window into per-stream big-number accumulators
//   k_stream_offsets + k_pack_streams : wavefront prefix scan of the stream byte lengths, coalesced packing
//   k_chunk_decode : one wavefront owns one stream: inclusive-cumulative table in registers (4 entries per lane),
//                    symbol search by wave-wide compare + ballot (no division by the data-dependent r)
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdexcept>
#include <type_traits>

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

constexpr int kWin = 72;   // LDS accumulation window in 32-bit words (64 symbols x <= 31 shifts = 62 words + 2)

// One wavefront owns one stream (arith::Encoder<uint32_t>, arith/coder.h:58-91, with an adaptive table of
// arith/stat_adaptive.h).  Per batch of 64 symbols:
//   * the model is evaluated by counting: symbol j of the batch sees the table at the batch start plus the symbols before it
//     inside the batch -- how many are smaller / equal comes from nine ballots (one per symbol bit, MSB first), no loop over lanes;
//   * the range register follows its serial recurrence on the scalar unit: r = floor(R / t) through the reciprocal of t
//     (t = t0 + position is known in advance: one s_mul_hi + shift), R' = r x or R - r x, renormalised with one
//     count-leading-zeros; about a dozen scalar instructions per symbol;
//   * the low register is a sum: L = sum r_k l_k 2^(-S_k).  Every lane adds its term into an LDS window at its bit position,
//     the window goes to the stream's accumulator words (one 64-bit counter per 32-bit output word, folded and carried by
//     k_carry_*): bit-plus-follow is carry propagation.
__global__ __launch_bounds__(64) void k_chunk_encode(const StreamJob *jobs, const uint32_t *inits, const MagicEnt *magic,
                                                     unsigned long long *acc, uint32_t *stream_bits)
{
	const StreamJob jb = jobs[blockIdx.x];
	const int lane = threadIdx.x;
	__shared__ uint32_t cnt[257], cum[257], bh[256];
	__shared__ unsigned long long win[kWin];
	{
		const uint32_t *st = inits + (size_t)jb.init * 256 + 4 * lane;
		uint32_t a = st[0], b = st[1], c = st[2], d = st[3], tot;
		uint32_t ex = wscan_excl(a + b + c + d, tot);
		cnt[4 * lane] = a; cnt[4 * lane + 1] = b; cnt[4 * lane + 2] = c; cnt[4 * lane + 3] = d;
		cum[4 * lane] = ex; cum[4 * lane + 1] = ex + a; cum[4 * lane + 2] = ex + a + b; cum[4 * lane + 3] = ex + a + b + c;
		if (lane == 0) { cnt[256] = 0; cum[256] = 0; }
	}
	__syncthreads();
	uint32_t R = 1u << 31;   // coder.h:47 with b = 32
	uint32_t S = 0;
	const uint64_t earlier = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	for (uint32_t base = 0; base < jb.n; base += 64) {
		const uint32_t j = base + lane;
		const bool valid = j < jb.n;
		const uint32_t nb = min(64u, jb.n - base);
		const uint32_t s = valid ? jb.sym[j] : 0x100u;   // the marker is larger than every symbol: never counted below a valid lane
		// lanes of the batch with a smaller / an equal symbol, bit by bit from the top
		uint64_t eq = ~0ull, lt = 0;
#pragma unroll
		for (int b = 8; b >= 0; --b) {
			const bool mine = (s >> b) & 1u;
			const uint64_t m = __ballot(mine);
			lt |= mine ? (eq & ~m) : 0ull;
			eq &= mine ? m : ~m;
		}
		const uint32_t l = cum[s] + (uint32_t)__popcll(lt & earlier), c = cnt[s] + (uint32_t)__popcll(eq & earlier);
		// per-symbol constants of the recurrence
		const uint32_t t = jb.t0 + j;
		const bool sub = l + c == t;   // the last symbol with a non-zero count: R' = R - r l (coder.h:74-77)
		MagicEnt me = valid ? magic[t] : MagicEnt{ 0, 0, 0 };
		const uint32_t mx = valid ? (sub ? l : c) : 0u;
		const uint32_t mm = ((me.shift >> kMagicSh32Shift) & 31u) | (sub || !valid ? 0x80u : 0u);
		// serial recurrence over the batch (coder.h:69-91 without the low register), uniform: lives in scalar registers
		uint32_t my_r = 0, my_s = 0;
		const uint32_t s_first = S;
		auto step = [&](uint32_t i) {
			const uint32_t M = (uint32_t)__builtin_amdgcn_readlane((int)me.m32, (int)i);
			const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)mx, (int)i);
			const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)mm, (int)i);
			const uint32_t r = __umulhi(R, M) >> (meta & 31u);
			const uint32_t prod = r * x;
			const uint32_t Rn = (meta & 0x80u) ? R - prod : prod;
			const uint32_t y = Rn - 1u;
			const uint32_t sh = y ? (uint32_t)__builtin_clz(y) - 1u : 31u;
			my_r = lane == (int)i ? r : my_r;
			my_s = lane == (int)i ? S : my_s;
			R = Rn << sh;
			S += sh;
		};
		// unrolled by hand: a taken branch costs a lone wavefront more than the few instructions of a step (the decoder's
		// step is four times as long: unrolling it measured slower)
		uint32_t i = 0;
		for (; i + 4 <= nb; i += 4) { step(i); step(i + 1); step(i + 2); step(i + 3); }
		for (; i < nb; ++i) step(i);
		// low register: L += r * l at bit position my_s (coder.h:71); r l <= R fits 32 bits
		const uint32_t w0 = s_first >> 5, span = ((S + 63) >> 5) - w0 + 1;   // <= 64 words
		const uint32_t a = (valid && l) ? my_r * l : 0u;
		const uint32_t w = my_s >> 5, shb = my_s & 31;
		const uint32_t hi = a >> shb, lo = shb ? a << (32 - shb) : 0u;
		unsigned long long *dst = acc + jb.word_base;
		for (uint32_t k = lane; k < span; k += 64) win[k] = 0;
		bh[4 * lane] = 0; bh[4 * lane + 1] = 0; bh[4 * lane + 2] = 0; bh[4 * lane + 3] = 0;
		__syncthreads();
		if (hi) atomicAdd(&win[w - w0], (unsigned long long)hi);
		if (lo) atomicAdd(&win[w - w0 + 1], (unsigned long long)lo);
		if (valid) atomicAdd(&bh[s], 1u);
		__syncthreads();
		for (uint32_t k = lane; k < span; k += 64) {
			unsigned long long v = win[k];
			if (v) atomicAdd(&dst[w0 + k], v);
		}
		// adaptive update of the tables by the whole batch (stat_adaptive.h:77-82)
		uint32_t a0 = bh[4 * lane], a1 = bh[4 * lane + 1], a2 = bh[4 * lane + 2], a3 = bh[4 * lane + 3], tot;
		uint32_t ex = wscan_excl(a0 + a1 + a2 + a3, tot);
		cnt[4 * lane] += a0; cnt[4 * lane + 1] += a1; cnt[4 * lane + 2] += a2; cnt[4 * lane + 3] += a3;
		cum[4 * lane] += ex; cum[4 * lane + 1] += ex + a0; cum[4 * lane + 2] += ex + a0 + a1; cum[4 * lane + 3] += ex + a0 + a1 + a2;
		__syncthreads();
	}
	if (lane == 0) stream_bits[blockIdx.x] = S + 32;   // flush: the 32 bits of the low register (coder.h:58-67)
}

// ---------------------------------------------------------------------------------------------------------
// The encoder in TWO kernels, for containers of thousands of streams (round 5).  k_chunk_encode above is bound by the ONE scalar
// unit of a compute unit: every symbol's range step is a dozen scalar instructions on a serial chain, and the chains of all the
// wavefronts of a compute unit share that unit (879 M symbols x ~18 / (256 units x 2.4 GHz) = 26 of the 28 ms the configs[3] mesh
// takes at the named size).  The model does not need the range register, and the range register does not need the wavefront:
//   k_chunk_model   a wavefront per stream, as above, but only the adaptive model by counting: per symbol (l, count) and floor(2^32 / t),
//                   8 bytes, stream after stream -- vector work, 64 symbols a step;
//   k_chunk_ranges  a LANE per stream, 64 streams per wavefront: arith::Encoder<uint32_t>::operator() (arith/coder.h:69-91) over
//                   the records -- r = floor(R / t) by the reciprocal of t (t = t0 + position), the interval, one count-leading-
//                   zeros for the renormalisation -- on the vector unit, sixty-four chains an instruction; the low register as
//                   before is a sum of terms r l 2^-S: a lane keeps the two 32-bit output words under its position as 64-bit
//                   counters and stores a word when its position has left it (k_carry_* folds and carries, unchanged).
// Same bits as k_chunk_encode, stream for stream (tests: the container against the oracle's at both sizes of the switch).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_chunk_model(const StreamJob *jobs, const uint32_t *inits, const MagicEnt *magic, const unsigned long long *rec_off, uint2 *rec)
{
	const StreamJob jb = jobs[blockIdx.x];
	const int lane = threadIdx.x;
	__shared__ uint32_t cnt[257], cum[257], bh[256];
	{
		const uint32_t *st = inits + (size_t)jb.init * 256 + 4 * lane;
		uint32_t a = st[0], b = st[1], c = st[2], d = st[3], tot;
		uint32_t ex = wscan_excl(a + b + c + d, tot);
		cnt[4 * lane] = a; cnt[4 * lane + 1] = b; cnt[4 * lane + 2] = c; cnt[4 * lane + 3] = d;
		cum[4 * lane] = ex; cum[4 * lane + 1] = ex + a; cum[4 * lane + 2] = ex + a + b; cum[4 * lane + 3] = ex + a + b + c;
		if (lane == 0) { cnt[256] = 0; cum[256] = 0; }
	}
	__syncthreads();
	uint2 *out = rec + rec_off[blockIdx.x];
	const uint64_t earlier = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	for (uint32_t base = 0; base < jb.n; base += 64) {
		const uint32_t j = base + lane;
		const bool valid = j < jb.n;
		const uint32_t s = valid ? jb.sym[j] : 0x100u;
		uint64_t eq = ~0ull, lt = 0;
#pragma unroll
		for (int b = 8; b >= 0; --b) {
			const bool mine = (s >> b) & 1u;
			const uint64_t m = __ballot(mine);
			lt |= mine ? (eq & ~m) : 0ull;
			eq &= mine ? m : ~m;
		}
		const uint32_t l = cum[s] + (uint32_t)__popcll(lt & earlier), c = cnt[s] + (uint32_t)__popcll(eq & earlier);
		if (valid) {
			// ... and floor(2^32 / t) for the range step (t = t0 + j; from the table's reciprocal floor(2^(32+s) / t) + 1 at shift s;
			// a power of two has 2^31 at shift s - 1)
			const MagicEnt me = magic[jb.t0 + j];
			const uint32_t msh = (me.shift >> kMagicSh32Shift) & 31u;
			const uint32_t q = me.m32 == 0x80000000u ? me.m32 >> msh : (me.m32 - 1u) >> msh;   // (2^31 >> (s - 1) = 2^(32 - s))
			out[j] = make_uint2(l | (c << 16), q);   // (l and c below t0 + n <= 65535: the launcher checks)
		}
		bh[4 * lane] = 0; bh[4 * lane + 1] = 0; bh[4 * lane + 2] = 0; bh[4 * lane + 3] = 0;
		__syncthreads();
		if (valid) atomicAdd(&bh[s], 1u);
		__syncthreads();
		uint32_t a0 = bh[4 * lane], a1 = bh[4 * lane + 1], a2 = bh[4 * lane + 2], a3 = bh[4 * lane + 3], tot;
		uint32_t ex = wscan_excl(a0 + a1 + a2 + a3, tot);
		cnt[4 * lane] += a0; cnt[4 * lane + 1] += a1; cnt[4 * lane + 2] += a2; cnt[4 * lane + 3] += a3;
		cum[4 * lane] += ex; cum[4 * lane + 1] += ex + a0; cum[4 * lane + 2] += ex + a0 + a1; cum[4 * lane + 3] += ex + a0 + a1 + a2;
		__syncthreads();
	}
}

// order: the streams longest first (waves of equally long chains); lane k of workgroup g runs stream order[64 g + k].
// A lane's records lie one after the other in memory -- a load per lane would be 64 cache lines a wavefront -- so the wavefront
// brings them in TOGETHER, a stream at a time: lane k loads record step0 + k of stream s (512 contiguous bytes), 64 such loads
// fill two 64 x 64 tiles in LDS, transposed on the way (row s, 65 words apart: both the write of a row and a lane's walk along
// its own row touch every bank once); the next tiles' loads are in flight while these tiles' 64 steps run.
// A lone wavefront issues an instruction every five or six cycles whatever it is, so a step is written for FEW instructions and
// no branches: floor(R / t) = umulhi(R, floor(2^32 / t)) or one more (the estimate is short by less than R / 2^32 <= 1/2: one
// compare of the remainder), both interval candidates computed and selected, the term's two words by one 64-bit shift.  What is
// left data-dependent is the store of an output word when a lane's position has left it.
// (Versions on the way: record and table reciprocal loaded per lane a step ahead: 580 cycles a step; eight steps ahead: the same
// -- the wavefront is bound by what it issues, not by what it waits for; the compiler's branches around the float-reciprocal
// division and the two products: 640.)
__global__ __launch_bounds__(64) void k_chunk_ranges(const StreamJob *jobs, uint32_t njobs, const uint32_t *order, const unsigned long long *rec_off,
                                                     const uint2 *rec, unsigned long long *acc, uint32_t *stream_bits)
{
	__shared__ uint32_t tile_lc[64 * 65], tile_q[64 * 65];
	const uint32_t lane = threadIdx.x;
	const uint32_t slot = blockIdx.x * 64 + lane;
	const bool have = slot < njobs;
	const uint32_t job = have ? order[slot] : 0u;
	StreamJob jb{ nullptr, 0, 0, 256, 0 };
	if (have) jb = jobs[job];
	const uint32_t n = have ? jb.n : 0u;
	const uint2 *in = rec + (have ? rec_off[job] : 0ull);
	unsigned long long *dst = acc + jb.word_base;
	uint32_t nmax = n;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d, 64));
	nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
	const uint32_t in_lo = (uint32_t)(uintptr_t)in, in_hi = (uint32_t)((uintptr_t)in >> 32);
	uint2 nxt[64];
	// the records step0 .. step0 + 63 of every stream of the wavefront, one coalesced load a stream
	auto fetch = [&](uint32_t step0) {
#pragma unroll
		for (int s = 0; s < 64; ++s) {
			const uint32_t ns = (uint32_t)__builtin_amdgcn_readlane((int)n, s);
			const uint2 *ps = (const uint2*)(((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)in_hi, s) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)in_lo, s));
			const uint32_t j = step0 + lane;
			nxt[s] = j < ns ? ps[j] : make_uint2(0u, 0u);
		}
	};
	auto stash = [&]() {
#pragma unroll
		for (int s = 0; s < 64; ++s) { tile_lc[s * 65 + lane] = nxt[s].x; tile_q[s * 65 + lane] = nxt[s].y; }
	};
	uint32_t R = 1u << 31, S = 0, t = jb.t0;   // coder.h:47 with b = 32
	uint32_t cur_w = 0;
	unsigned long long a0 = 0, a1 = 0;   // the output words cur_w and cur_w + 1 as counters
	fetch(0);
	stash();
	for (uint32_t step0 = 0; step0 < nmax; step0 += 64) {
		const bool more = step0 + 64 < nmax;
		if (more) fetch(step0 + 64);
		const uint32_t *row_lc = tile_lc + lane * 65, *row_q = tile_q + lane * 65;
		const uint32_t left = nmax - step0 < 64u ? nmax - step0 : 64u;   // (uniform)
		for (uint32_t k = 0; k < left; ++k) {
			const uint32_t lc = row_lc[k], q = row_q[k];
			const bool live = step0 + k < n;   // (a lane past its stream's end goes through the motions on zeros and keeps its state)
			const uint32_t l = lc & 0xffffu, c = lc >> 16;
			uint32_t r = __umulhi(R, q);
			r += R - r * t >= t ? 1u : 0u;      // floor(R / t)
			uint32_t rl, rc;
			asm("v_mul_lo_u32 %0, %1, %2" : "=v"(rl) : "v"(r), "v"(l));   // (both products, unconditionally: the compiler branches around one of them)
			asm("v_mul_lo_u32 %0, %1, %2" : "=v"(rc) : "v"(r), "v"(c));
			const uint32_t Rn = l + c == t ? R - rl : rc;   // the last symbol with a count: R' = R - r l (coder.h:74-77)
			// low register: L += r l at bit position S (coder.h:71); r l <= R fits 32 bits: its two words by one 64-bit shift
			const uint32_t w = S >> 5;
			const unsigned long long two = ((unsigned long long)(live ? rl : 0u) << 32) >> (S & 31u);
			if (w != cur_w) { dst[cur_w] = a0; a0 = a1; a1 = 0; cur_w = w; }   // (a step moves the position by at most 31 bits: one word)
			a0 += two >> 32;
			a1 += two & 0xffffffffull;
			const uint32_t y = Rn - 1u;
			const uint32_t sh = y ? (uint32_t)__builtin_clz(y) - 1u : 31u;
			R = live ? Rn << sh : R;
			S += live ? sh : 0u;
			t += live ? 1u : 0u;
		}
		if (more) stash();
	}
	if (have) {
		dst[cur_w] = a0;
		dst[cur_w + 1u] = a1;
		stream_bits[job] = S + 32u;   // flush: the 32 bits of the low register (coder.h:58-67)
	}
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
// decode: arith::Decoder<uint32_t> (coder.h:115-162) + stat_adaptive.h:55-72.  target = min(t - 1, D / r) and the Fenwick
// descent are replaced by their definition: the symbol is the number of table entries whose inclusive cumulative count I
// satisfies I <= target  <=>  I * r <= D and I < t.  Each lane holds 4 consecutive entries in registers; I * r <= R <= 2^31
// never overflows.  Entries with I == t are exactly those from the last symbol with a non-zero count on (a count never
// becomes non-zero later), so "and I < t" is a clamp to that symbol, fixed per stream.  The stream's bits reach the
// wavefront 2048 at a time: lane k holds big-endian word k of the window that starts at the batch's byte position (a batch
// consumes at most 64 x 31 bits), and the bits a renormalisation shifts in are cut out of two neighbouring words.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_chunk_decode(const StreamJob *jobs, const uint32_t *inits, const MagicEnt *magic,
                                                     const uint8_t *payload, const unsigned long long *offsets, const uint32_t *nbytes, uint8_t *sym_out_base)
{
	const StreamJob jb = jobs[blockIdx.x];
	const int lane = threadIdx.x;
	uint8_t *out = const_cast<uint8_t*>(jb.sym);
	(void)sym_out_base;
	// inclusive cumulative counts of entries 4*lane .. 4*lane+3
	uint32_t i0, i1, i2, i3, sym_last;
	{
		const uint32_t *st = inits + (size_t)jb.init * 256 + 4 * lane;
		const uint32_t c0 = st[0], c1 = st[1], c2 = st[2], c3 = st[3];
		uint32_t tot, ex = wscan_excl(c0 + c1 + c2 + c3, tot);
		i0 = ex + c0; i1 = i0 + c1; i2 = i1 + c2; i3 = i2 + c3;
		const uint32_t top = c3 ? 4 * lane + 3 : c2 ? 4 * lane + 2 : c1 ? 4 * lane + 1 : c0 ? 4 * lane : 0u;
		uint32_t mx = top;   // largest symbol with a non-zero count
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
		sym_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)mx);
	}
	const uint8_t *src = payload + offsets[blockIdx.x];
	const uint32_t nby = nbytes[blockIdx.x];
	auto byte_at = [&](uint32_t k) -> uint32_t { return k < nby ? src[k] : 0xffu; };   // reads past the end give 0xFF (bitstream.h:27)
	uint32_t D = (byte_at(0) << 24) | (byte_at(1) << 16) | (byte_at(2) << 8) | byte_at(3);   // coder.h:124-129
	D = (uint32_t)__builtin_amdgcn_readfirstlane((int)D);
	uint32_t pos = 32;   // stream bits consumed so far
	uint32_t R = 1u << 31;
	for (uint32_t base = 0; base < jb.n; base += 64) {
		const uint32_t nb = min(64u, jb.n - base);
		// reciprocals of the totals of the next 64 symbols: t is known in advance (t0 + position)
		const MagicEnt me = base + lane < jb.n ? magic[jb.t0 + base + lane] : MagicEnt{ 0, 0, 0 };
		const uint32_t msh = (me.shift >> kMagicSh32Shift) & 31u;
		// bit window of this batch
		const uint32_t wbyte = pos >> 3;
		const uint32_t kb = wbyte + 4 * lane;
		uint32_t wword;
		if (kb + 4 <= nby) { uint32_t raw; __builtin_memcpy(&raw, src + kb, 4); wword = __builtin_bswap32(raw); }
		else wword = (byte_at(kb) << 24) | (byte_at(kb + 1) << 16) | (byte_at(kb + 2) << 8) | byte_at(kb + 3);
		uint32_t wo = pos & 7u;   // bit offset of the next unread bit inside the window
		uint32_t mysym = 0;
		// I * r <= R <= 2^31 always; once t > 128, r = floor(R / t) < 2^24 and I <= t < 2^24: the 24-bit multiplier (full rate)
		// gives the exact product
		auto run = [&](auto use24) {
			for (uint32_t i = 0; i < nb; ++i) {
				const uint32_t t = jb.t0 + base + i;
				const uint32_t M = (uint32_t)__builtin_amdgcn_readlane((int)me.m32, (int)i);
				const uint32_t shf = (uint32_t)__builtin_amdgcn_readlane((int)msh, (int)i);
				const uint32_t r = __umulhi(R, M) >> shf;
				// symbol = min(#{ entries : I * r <= D }, last symbol with a non-zero count)
				uint32_t p0, p1, p2, p3;
				if constexpr (decltype(use24)::value) { p0 = __umul24(i0, r); p1 = __umul24(i1, r); p2 = __umul24(i2, r); p3 = __umul24(i3, r); }
				else { p0 = i0 * r; p1 = i1 * r; p2 = i2 * r; p3 = i3 * r; }
				const uint64_t b0 = __ballot(p0 <= D), b1 = __ballot(p1 <= D), b2 = __ballot(p2 <= D), b3 = __ballot(p3 <= D);
				const uint32_t below = (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
				const uint32_t s = min(below, sym_last);
				// r l and r h are the products of the entries s - 1 and s
				const uint32_t ls = s >> 2, k = s & 3u;
				const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)p0, (int)ls), e1 = (uint32_t)__builtin_amdgcn_readlane((int)p1, (int)ls);
				const uint32_t e2 = (uint32_t)__builtin_amdgcn_readlane((int)p2, (int)ls), e3 = (uint32_t)__builtin_amdgcn_readlane((int)p3, (int)ls);
				const uint32_t pv = (uint32_t)__builtin_amdgcn_readlane((int)p3, (int)(ls ? ls - 1 : 0));
				const uint32_t rh = k == 0 ? e0 : k == 1 ? e1 : k == 2 ? e2 : e3;
				const uint32_t rl = k == 0 ? (ls ? pv : 0u) : k == 1 ? e0 : k == 2 ? e1 : e2;
				// coder.h:140-153; h < t  <=>  s is not the last symbol with a non-zero count
				const uint32_t Rn = s < sym_last ? rh - rl : R - rl;
				const uint32_t y = Rn - 1u;
				const uint32_t sh = y ? (uint32_t)__builtin_clz(y) - 1u : 31u;
				R = Rn << sh;
				// shift in the next sh bits of the stream
				const uint32_t wk = wo >> 5, wb = wo & 31u;
				const uint64_t two = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wword, (int)wk) << 32) |
				                     (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wword, (int)min(wk + 1u, 63u));
				const uint32_t bits = (uint32_t)(((two << wb) >> 1) >> (63u - sh));
				D = ((D - rl) << sh) | bits;
				wo += sh;
				// adaptive update (stat_adaptive.h:77-82): inclusive counts of every entry >= s
				const uint32_t e = 4 * lane;
				i0 += (e >= s); i1 += (e + 1 >= s); i2 += (e + 2 >= s); i3 += (e + 3 >= s);
				mysym = lane == (int)i ? s : mysym;
				(void)t;
			}
		};
		if (jb.t0 + base > 128u) run(std::true_type()); else run(std::false_type());
		pos = 8 * wbyte + wo;
		if (base + lane < jb.n) out[base + lane] = (uint8_t)mysym;
	}
}

// ---------------------------------------------------------------------------------------------------------
// The same decoder for the many-streams regime: ONE LANE per stream, 64 streams per wavefront (round 4).
// k_chunk_decode spends a wavefront on a stream: 48 scalar + 28 vector instructions per symbol, and once there are eight waves
// per SIMD the scalar port is the bound (configs[3]: 7.9 G symbols/s, 111 ms for the attribute planes of 100 M triangles).
// Here a lane runs the reference's loop on its own stream (arith::Decoder<uint32_t>::decode, coder.h:134-153, with
// stat_adaptive.h:55-82), so a wave instruction serves 64 symbols:
//   * the table has two levels: 16 inclusive block sums in registers and the 256 COUNTS in LDS, entry e of lane l at word
//     e * 64 + l (bank = lane: no conflicts); the target min(t - 1, D / r) is looked up by two scans of 16 (first the block,
//     then the symbol in it by a running sum) -- the Fenwick descent of the reference (stat_adaptive.h:55-72) flattened to
//     two levels -- and the update is one add to the count and one to every block sum from its block on;
//   * D / r: r = floor(R / t) < 2^24 once t > 128 and the quotient stays below 2^23, so a float reciprocal lands within two
//     of it and four compare-and-adjust steps make it exact; r l and r (h - l) are 24-bit multiplies (<= R < 2^32);
//   * the stream's bits: a 64-bit buffer per lane, topped up by four bytes whenever fewer than 32 bits are left (a
//     renormalisation takes at most 31); bytes past the end read 0xFF (bitstream.h:27).
// Streams whose table starts with t0 <= 128 (planes under 1024 symbols keep the reference's initial counts) stay with
// k_chunk_decode: the host sorts them behind the others.  About 2.5 vector instructions per symbol instead of 76.
// ---------------------------------------------------------------------------------------------------------
// CT: the counts' type in LDS.  uint16_t where every stream of the launch has t0 + n <= 65535 (a count never exceeds the total): 32 KB
// a wavefront instead of 64, five wavefronts a compute unit instead of two -- the 840 lane-wavefronts of the 100 M-triangle mesh's
// connectivity streams (and the 830 of its attribute streams) in ONE round instead of two (round 5).  Entry e of lane l at element
// e * 64 + l either way: two lanes' 16-bit counts share a bank's word, which is a broadcast, not a conflict.
template <typename CT>
__global__ __launch_bounds__(64) void k_chunk_decode_lanes(const StreamJob *jobs, uint32_t njobs, const uint32_t *inits, const MagicEnt *magic,
                                                           const uint8_t *payload, const unsigned long long *offsets, const uint32_t *nbytes)
{
	extern __shared__ uint32_t lds_tab[];
	CT *const cnt = (CT*)lds_tab;                // [256][64]
	const uint32_t lane = threadIdx.x;
	const uint32_t j = blockIdx.x * 64 + lane;
	const bool have = j < njobs;
	StreamJob jb{ nullptr, 0, 0, 256, 0 };
	if (have) jb = jobs[j];
	// the block level lives in registers as inclusive cumulative sums (no trip to LDS for it: with one wave per SIMD -- a few
	// thousand streams are a hundred waves -- every trip is waited for in full)
	uint32_t C[16];
	{
		const uint32_t *st = inits + (size_t)jb.init * 256;
		uint32_t sum = 0;
#pragma unroll
		for (int b = 0; b < 16; ++b) {
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				const uint32_t c = have ? st[b * 16 + k] : 0u;
				cnt[(b * 16 + k) * 64 + lane] = (CT)c;
				sum += c;
			}
			C[b] = sum;
		}
	}
	const uint8_t *src = payload + (have ? offsets[j] : 0ull);
	const uint32_t nby = have ? nbytes[j] : 0u;
	auto byte_at = [&](uint32_t k) -> uint32_t { return k < nby ? src[k] : 0xffu; };
	auto word_at = [&](uint32_t k) -> uint32_t {
		if (k + 4 <= nby) { uint32_t raw; __builtin_memcpy(&raw, src + k, 4); return __builtin_bswap32(raw); }
		return (byte_at(k) << 24) | (byte_at(k + 1) << 16) | (byte_at(k + 2) << 8) | byte_at(k + 3);
	};
	uint32_t D = word_at(0);   // coder.h:124-129
	uint32_t p = 8;            // first byte behind the word that waits in `ahead`
	uint32_t ahead = word_at(4);   // the stream's next four bytes, asked for a refill early: the load is not waited for when it is needed
	uint64_t buf = 0;          // unread bits, left-aligned
	uint32_t avail = 0;
	uint32_t R = 1u << 31, t = jb.t0;
	typedef __attribute__((address_space(1))) uint8_t gbyte;   // (global, not generic: a flat store also counts as an LDS operation)
	gbyte *out = (gbyte*)const_cast<uint8_t*>(jb.sym);
	uint32_t nmax = jb.n;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d, 64));
	nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
	MagicEnt me = magic[t];    // reciprocal of t, fetched a symbol ahead (t is known: t0 + position)
	// decoded symbols leave sixteen at a time: a store per symbol has every later wait for a load wait for the store as well (one
	// counter for both, in order) -- a round trip to memory per symbol, 124 ms for the configs[3] mesh's attribute planes instead of 30
	uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;   // the last sixteen symbols, oldest in the lowest byte of o0
	for (uint32_t i = 0; i < nmax; ++i) {
		if ((i & 15u) == 0u && i) {   // (uniform: every lane stores the sixteen symbols before i, if it has that many)
			if (i < jb.n) {   // (a stream that ends exactly here has stored its last sixteen itself, below)
				typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
				typedef __attribute__((address_space(1))) u32x4 __attribute__((aligned(1))) gvec;
				*(gvec*)(out + i - 16) = u32x4{ o0, o1, o2, o3 };
			}
		}
		if (i >= jb.n) continue;
		const uint32_t r = __umulhi(R, me.m32) >> ((me.shift >> kMagicSh32Shift) & 31u);   // floor(R / t)
		me = magic[t + 1u];
		// target = min(t - 1, D / r)
		uint32_t q = (uint32_t)((float)D * __builtin_amdgcn_rcpf((float)r));
		int32_t rem = (int32_t)(D - __umul24(q, r));
		if (rem < 0) { --q; rem += (int32_t)r; }
		if (rem < 0) { --q; rem += (int32_t)r; }
		if (rem >= (int32_t)r) { ++q; rem -= (int32_t)r; }
		if (rem >= (int32_t)r) { ++q; rem -= (int32_t)r; }
		const uint32_t target = min(t - 1u, q);
		// the block: how many block sums stay at or below the target, and the count below that block
		uint32_t b = 0, lo = 0;
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			const bool le = C[k] <= target;
			b += le ? 1u : 0u;
			lo = le ? C[k] : lo;
		}
		// the symbol inside the block (target < t = the sum of all counts, so the block exists): the counts before it that fit,
		// and the first running sum that does not (its difference to the last one that does is the symbol's count)
		const CT *cb = cnt + (size_t)(b * 16u) * 64 + lane;
		uint32_t sin = 0, run = lo, hi = 0xffffffffu;
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			run += (uint32_t)cb[k * 64];
			const bool le = run <= target;
			sin += le ? 1u : 0u;
			lo = le ? run : lo;
			hi = min(hi, le ? 0xffffffffu : run);
		}
		const uint32_t s = b * 16u + sin;
		const uint32_t cs = hi - lo;
		// coder.h:140-153 with l = lo, h = hi
		const uint32_t rl = __umul24(r, lo);
		const uint32_t Rn = hi < t ? __umul24(r, cs) : R - rl;
		const uint32_t y = Rn - 1u;
		const uint32_t sh = y ? (uint32_t)__builtin_clz(y) - 1u : 31u;
		R = Rn << sh;
		if (avail < 32u) { buf |= (uint64_t)ahead << (32u - avail); avail += 32u; ahead = word_at(p); p += 4u; }
		const uint32_t bits = (uint32_t)((buf >> 1) >> (63u - sh));
		buf <<= sh; avail -= sh;
		D = ((D - rl) << sh) | bits;
		// stat_adaptive.h:77-82
		cnt[s * 64 + lane] = (CT)(cs + 1u);
#pragma unroll
		for (int k = 0; k < 16; ++k) C[k] += (uint32_t)k >= b ? 1u : 0u;
		++t;
		o0 = __builtin_amdgcn_alignbyte(o1, o0, 1); o1 = __builtin_amdgcn_alignbyte(o2, o1, 1); o2 = __builtin_amdgcn_alignbyte(o3, o2, 1);
		o3 = (o3 >> 8) | (s << 24);
		if (i + 1u == jb.n) {   // the stream's last symbols: the ones no store of sixteen will take
			const uint32_t m = jb.n & 15u ? jb.n & 15u : 16u;
			auto shift = [&] { o0 = __builtin_amdgcn_alignbyte(o1, o0, 1); o1 = __builtin_amdgcn_alignbyte(o2, o1, 1); o2 = __builtin_amdgcn_alignbyte(o3, o2, 1); o3 >>= 8; };
			for (uint32_t k = m; k < 16u; ++k) shift();
			for (uint32_t k = 0; k < m; ++k) { out[jb.n - m + k] = (uint8_t)o0; shift(); }
		}
	}
}

// (index, value) pairs -> dst[index] = value (late twin links of the pipelined decode; the pairs are applied in order of
// appearance only across launches -- inside one list an index occurs at most once per link, later pairs repeat the final value)
__global__ __launch_bounds__(256) void k_scatter_u32(const uint32_t *pairs, uint32_t n, uint32_t *dst)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[pairs[2 * i]] = pairs[2 * i + 1];
}

// 256-bin histogram of every plane (static priors): one workgroup per 16 Ki-symbol slice, LDS bins, one global add per bin
__global__ __launch_bounds__(256) void k_plane_hist(const HistSlice *slices, uint32_t *hist)
{
	const HistSlice sl = slices[blockIdx.x];
	__shared__ uint32_t bins[256];
	bins[threadIdx.x] = 0;
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < sl.n; i += 256) atomicAdd(&bins[sl.sym[i]], 1u);
	__syncthreads();
	if (bins[threadIdx.x]) atomicAdd(&hist[(size_t)sl.plane * 256 + threadIdx.x], bins[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// operation planes: the walk emits one byte per cut-border operation (symbol | order class << 3, models.h:101-105); the
// container codes one plane per class.  Stable partition in three small launches: a wavefront counts the classes of its unit
// of kSplitUnit operations, one wavefront scans the units per class, then every wavefront places its unit: rank inside a
// 64-operation step = population count of the ballot below the lane.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_split_count(const uint8_t *ops, uint32_t n, uint32_t *cnt)
{
	const uint32_t u = blockIdx.x, lane = threadIdx.x;
	uint32_t c[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	for (uint32_t i = u * kSplitUnit + lane; i < min(n, (u + 1) * kSplitUnit); i += 64) {
		const uint32_t k = ops[i] >> 3;
#pragma unroll
		for (int j = 0; j < 8; ++j) c[j] += k == (uint32_t)j;
	}
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		uint32_t v = c[j];
		for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
		if (lane == 0) cnt[u * 8 + j] = v;
	}
}
__global__ __launch_bounds__(64) void k_split_scan(uint32_t *cnt, uint32_t nunits, SplitBase base)
{
	// one wavefront per class: exclusive scan of the class's counts over the units, 64 units per step
	const uint32_t j = blockIdx.x, lane = threadIdx.x;
	uint32_t run = base.b[j];
	for (uint32_t u0 = 0; u0 < nunits; u0 += 64) {
		const uint32_t u = u0 + lane;
		const uint32_t c = u < nunits ? cnt[u * 8 + j] : 0u;
		uint32_t inc = c;
		for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if ((int)lane >= d) inc += o; }
		if (u < nunits) cnt[u * 8 + j] = run + inc - c;
		run += __shfl(inc, 63);
	}
}
__global__ __launch_bounds__(64) void k_split_place(const uint8_t *ops, uint32_t n, const uint32_t *cnt, uint8_t *planes)
{
	const uint32_t u = blockIdx.x, lane = threadIdx.x;
	uint32_t at[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) at[j] = cnt[u * 8 + j];
	const uint32_t end = min(n, (u + 1) * kSplitUnit);
	for (uint32_t i0 = u * kSplitUnit; i0 < end; i0 += 64) {
		const uint32_t i = i0 + lane;
		const bool live = i < end;
		const uint32_t b = live ? ops[i] : 0xffu, k = b >> 3;
		uint32_t dst = 0;
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const uint64_t mk = __ballot(live && k == (uint32_t)j);
			if (live && k == (uint32_t)j) dst = at[j] + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
			at[j] += (uint32_t)__popcll(mk);
		}
		if (live) planes[dst] = (uint8_t)(b & 7u);
	}
}
void launch_split_classes(hipStream_t st, const uint8_t *ops, uint32_t n, const uint32_t base[8], uint32_t *scratch, uint8_t *planes)
{
	if (!n) return;
	const uint32_t nunits = (n + kSplitUnit - 1) / kSplitUnit;
	SplitBase sb;
	for (int j = 0; j < 8; ++j) sb.b[j] = base[j];
	hipLaunchKernelGGL(k_split_count, dim3(nunits), dim3(64), 0, st, ops, n, scratch);
	hipLaunchKernelGGL(k_split_scan, dim3(8), dim3(64), 0, st, scratch, nunits, sb);
	hipLaunchKernelGGL(k_split_place, dim3(nunits), dim3(64), 0, st, ops, n, scratch, planes);
}

void launch_plane_hist(hipStream_t st, const HistSlice *slices, uint32_t nslices, uint32_t *hist)
{
	if (nslices) hipLaunchKernelGGL(k_plane_hist, dim3(nslices), dim3(256), 0, st, slices, hist);
}
void launch_scatter_u32(hipStream_t st, const uint32_t *pairs, uint32_t n, uint32_t *dst)
{
	if (n) hipLaunchKernelGGL(k_scatter_u32, dim3((n + 255) / 256), dim3(256), 0, st, pairs, n, dst);
}
void launch_chunk_encode(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *inits, const MagicEnt *magic, uint64_t *acc, uint32_t *stream_bits)
{
	if (nstreams) hipLaunchKernelGGL(k_chunk_encode, dim3(nstreams), dim3(64), 0, st, jobs, inits, magic, (unsigned long long*)acc, stream_bits);
}
// the same streams by k_chunk_model + k_chunk_ranges: rec_off[j] = records of the streams before j, rec = 8 bytes per symbol of all
// streams (the caller has checked t0 + n <= 65535 for every stream), order = the streams longest first
void launch_chunk_encode_split(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *inits, const MagicEnt *magic, uint64_t *acc, uint32_t *stream_bits,
                               const uint64_t *rec_off, void *rec, const uint32_t *order)
{
	if (!nstreams) return;
	hipLaunchKernelGGL(k_chunk_model, dim3(nstreams), dim3(64), 0, st, jobs, inits, magic, (const unsigned long long*)rec_off, (uint2*)rec);
	hipLaunchKernelGGL(k_chunk_ranges, dim3((nstreams + 63) / 64), dim3(64), 0, st, jobs, nstreams, order, (const unsigned long long*)rec_off, (const uint2*)rec,
	                   (unsigned long long*)acc, stream_bits);
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
// a lane per stream (every job: t0 > 128): 64 streams per workgroup, 64 KB of counts in LDS -- 32 KB with 16-bit counts
// (counts16: the caller has checked t0 + n <= 65535 for every stream of the launch)
template <typename CT>
static void launch_chunk_decode_lanes_as(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *inits, const MagicEnt *magic,
                                         const uint8_t *payload, const uint64_t *offsets, const uint32_t *nbytes)
{
	constexpr uint32_t kLds = 256 * 64 * sizeof(CT);
	// the dynamic LDS limit is a property of the function ON A DEVICE: raised once for every device a launch goes to (N contexts
	// on N devices in one process: hry_decode_sharded), on the device that is current -- the stream's
	static std::mutex mu;
	static bool raised[64] = {};
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
	{
		std::lock_guard<std::mutex> g(mu);
		if (!raised[dev]) {
			if (hipFuncSetAttribute((const void*)k_chunk_decode_lanes<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds) != hipSuccess) {
				(void)hipGetLastError();
				throw std::runtime_error("k_chunk_decode_lanes: the device refuses its dynamic LDS");
			}
			raised[dev] = true;
		}
	}
	hipLaunchKernelGGL(k_chunk_decode_lanes<CT>, dim3((nstreams + 63) / 64), dim3(64), kLds, st, jobs, nstreams, inits, magic, payload, (const unsigned long long*)offsets, nbytes);
	if (hipGetLastError() != hipSuccess) throw std::runtime_error("k_chunk_decode_lanes: launch failed");
}
void launch_chunk_decode_lanes(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *inits, const MagicEnt *magic,
                               const uint8_t *payload, const uint64_t *offsets, const uint32_t *nbytes, bool counts16)
{
	if (!nstreams) return;
	if (counts16) launch_chunk_decode_lanes_as<uint16_t>(st, jobs, nstreams, inits, magic, payload, offsets, nbytes);
	else launch_chunk_decode_lanes_as<uint32_t>(st, jobs, nstreams, inits, magic, payload, offsets, nbytes);
}

}   // namespace dev
}   // namespace hry
