// Hardware counters of the calling thread around a region (Linux perf_event_open, no external tool): cycles, instructions,
// branch misses, last-level-cache misses, L1D load misses.  Used under HRY_TRACE to put numbers behind "what bounds the two
// sequential host loops" (cut-border walk, replay).  Silently unavailable where the kernel forbids it (perf_event_paranoid).
#pragma once
#include <linux/perf_event.h>
#include <sys/ioctl.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

namespace hry {

struct PerfCounters {
	enum { N = 5 };
	int fd[N];
	uint64_t val[N];
	bool ok = false;
	static int open_one(uint32_t type, uint64_t config, int group)
	{
		perf_event_attr pe;
		memset(&pe, 0, sizeof pe);
		pe.type = type; pe.size = sizeof pe; pe.config = config;
		pe.disabled = group == -1 ? 1 : 0; pe.exclude_kernel = 1; pe.exclude_hv = 1;
		return (int)syscall(SYS_perf_event_open, &pe, 0, -1, group, 0);
	}
	// nothing is opened before start(): the first perf_event_open of a process costs 100 - 200 ms on a 256-thread host, and the
	// replay loops hold one of these whether or not HRY_PERF asks for the counters (round 3: that was the "cold first decode")
	PerfCounters() { for (int i = 0; i < N; ++i) { fd[i] = -1; val[i] = 0; } }
	void open()
	{
		if (ok || fd[0] >= 0) return;
		fd[0] = open_one(PERF_TYPE_HARDWARE, PERF_COUNT_HW_CPU_CYCLES, -1);
		if (fd[0] < 0) return;
		fd[1] = open_one(PERF_TYPE_HARDWARE, PERF_COUNT_HW_INSTRUCTIONS, fd[0]);
		fd[2] = open_one(PERF_TYPE_HARDWARE, PERF_COUNT_HW_BRANCH_MISSES, fd[0]);
		fd[3] = open_one(PERF_TYPE_HARDWARE, PERF_COUNT_HW_CACHE_MISSES, fd[0]);
		fd[4] = open_one(PERF_TYPE_HW_CACHE, PERF_COUNT_HW_CACHE_L1D | (PERF_COUNT_HW_CACHE_OP_READ << 8) | (PERF_COUNT_HW_CACHE_RESULT_MISS << 16), fd[0]);
		ok = true;
	}
	~PerfCounters() { for (int i = 0; i < N; ++i) if (fd[i] >= 0) close(fd[i]); }
	void start() { open(); if (ok) { ioctl(fd[0], PERF_EVENT_IOC_RESET, PERF_IOC_FLAG_GROUP); ioctl(fd[0], PERF_EVENT_IOC_ENABLE, PERF_IOC_FLAG_GROUP); } }
	void stop()
	{
		if (!ok) return;
		ioctl(fd[0], PERF_EVENT_IOC_DISABLE, PERF_IOC_FLAG_GROUP);
		for (int i = 0; i < N; ++i) { val[i] = 0; if (fd[i] >= 0 && read(fd[i], &val[i], 8) != 8) val[i] = 0; }
	}
	void report(const char *what, double per) const
	{
		if (!ok) { fprintf(stderr, "[hry perf] %s: hardware counters not available here (perf_event_open refused)\n", what); return; }
		fprintf(stderr, "[hry perf] %s: per triangle %.1f cycles, %.1f instructions (IPC %.2f), %.3f branch misses, %.3f LLC misses, %.3f L1D load misses\n",
		        what, val[0] / per, val[1] / per, val[0] ? (double)val[1] / val[0] : 0.0, val[2] / per, val[3] / per, val[4] / per);
	}
};

}   // namespace hry
