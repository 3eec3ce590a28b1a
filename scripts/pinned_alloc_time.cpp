// How long does pinned host memory take to come (hipHostMalloc) and to be written the first time, against pageable memory,
// and what does registering an existing block cost?  hipcc scripts/pinned_alloc_time.cpp -o /tmp/pat && /tmp/pat [MB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef std::chrono::steady_clock Clock;
static double ms(Clock::time_point t) { return std::chrono::duration<double, std::milli>(Clock::now() - t).count(); }
int main(int argc, char **argv)
{
	const size_t mb = argc > 1 ? atoi(argv[1]) : 512, n = mb << 20;
	(void)hipSetDevice(0);
	(void)hipFree(nullptr);
	void *d = nullptr;
	(void)hipMalloc(&d, n);
	for (int pass = 0; pass < 2; ++pass) {
		auto t = Clock::now();
		void *p = nullptr;
		if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
		const double t_alloc = ms(t);
		t = Clock::now(); memset(p, 1, n); const double t_touch = ms(t);
		t = Clock::now(); memset(p, 2, n); const double t_again = ms(t);
		t = Clock::now(); (void)hipMemcpy(d, p, n, hipMemcpyHostToDevice); const double t_copy = ms(t);
		t = Clock::now(); (void)hipHostFree(p); const double t_free = ms(t);
		printf("pinned   %zu MB: alloc %.1f ms, first write %.1f ms, second write %.1f ms, H2D %.1f ms (%.1f GB/s), free %.1f ms\n", mb, t_alloc, t_touch, t_again, t_copy, n / t_copy / 1e6, t_free);
	}
	for (int pass = 0; pass < 2; ++pass) {
		auto t = Clock::now();
		void *p = aligned_alloc(2 << 20, n);
		memset(p, 1, n);
		const double t_touch = ms(t);
		t = Clock::now(); (void)hipMemcpy(d, p, n, hipMemcpyHostToDevice); const double t_copy = ms(t);
		t = Clock::now();
		const hipError_t e = hipHostRegister(p, n, hipHostRegisterPortable);
		const double t_reg = ms(t);
		t = Clock::now(); (void)hipMemcpy(d, p, n, hipMemcpyHostToDevice); const double t_copy2 = ms(t);
		t = Clock::now(); (void)hipHostUnregister(p); const double t_unreg = ms(t);
		free(p);
		printf("pageable %zu MB: alloc + first write %.1f ms, H2D %.1f ms (%.1f GB/s); register %.1f ms (%s), H2D registered %.1f ms (%.1f GB/s), unregister %.1f ms\n", mb, t_touch, t_copy,
		       n / t_copy / 1e6, t_reg, hipGetErrorString(e), t_copy2, n / t_copy2 / 1e6, t_unreg);
	}
	return 0;
}
