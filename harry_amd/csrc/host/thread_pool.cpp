// Helper threads for parallel_for (host.hpp), kept between calls.
//
// parallel_for used to create its threads and join them every time: 0.7 - 1.5 ms for 32 - 64 threads on the 256-thread hosts
// (scripts/spawn_cost.cpp), and the analysis of a multi-component mesh calls it a dozen times in a row (labels, order, vertex
// bases, walks, concatenation), the shard planner and the replay some more -- 10 ms of a 56 ms host walk of the 12.6 M-triangle
// share of configs[3].  Now every CALLING thread owns a set of parked helpers (thread-local: the worker threads of the
// in-process N-device executor each have theirs, sized by their own thread budget); a call publishes the job under the set's
// mutex, runs index 0 itself and waits for the others.  Helpers are detached and hold the set's state alive; they leave when
// their owner thread ends (or with the process); a forked child starts without any.  A call made while the caller's set is busy (a body that calls parallel_for
// itself on the calling thread) falls back to threads of its own.
#include <pthread.h>
#include <sched.h>

#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include "host.hpp"

namespace hry {
namespace {
struct PoolState {
	std::mutex mu;
	std::condition_variable cv_work, cv_done;
	uint64_t gen = 0;
	unsigned n_job = 0;                      // indices 1 .. n_job - 1 take part in generation `gen`
	void (*fn)(void*, unsigned) = nullptr;
	void *arg = nullptr;
	const void *node = nullptr;
	unsigned pending = 0;
	unsigned n_helpers = 0;                  // created so far: indices 1 .. n_helpers
	bool closing = false, busy = false;
};
struct Pool {
	std::shared_ptr<PoolState> st;
	~Pool()
	{
		if (!st) return;
		{ std::lock_guard<std::mutex> g(st->mu); st->closing = true; }
		st->cv_work.notify_all();
	}
};
thread_local Pool t_pool;
// a forked child has the forking thread only: its copy of the set names helpers that do not exist there
void forget_helpers_in_child() { new (&t_pool.st) std::shared_ptr<PoolState>(); }   // (the old state is left alone: its mutex may be held)
struct AtFork { AtFork() { (void)pthread_atfork(nullptr, nullptr, forget_helpers_in_child); } };

thread_local bool t_is_helper = false;

void helper_main(std::shared_ptr<PoolState> st, unsigned index, uint64_t seen)
{
	t_is_helper = true;
	const void *bound = nullptr;
	bool bound_any = false;
	for (;;) {
		void (*fn)(void*, unsigned);
		void *arg;
		const void *node;
		{
			std::unique_lock<std::mutex> lk(st->mu);
			st->cv_work.wait(lk, [&] { return st->closing || st->gen != seen; });
			if (st->closing) return;
			seen = st->gen;
			if (index >= st->n_job) continue;
			fn = st->fn; arg = st->arg; node = st->node;
		}
		if (!bound_any || node != bound) { stay_on_node(node); bound = node; bound_any = true; }
		fn(arg, index);
		{
			std::lock_guard<std::mutex> lk(st->mu);
			if (--st->pending == 0) st->cv_done.notify_one();
		}
	}
}

// (a thread that cannot be created -- EAGAIN on a host at its limit -- leaves its index to the caller)
void run_on_fresh_threads(unsigned n, void (*fn)(void*, unsigned), void *arg, const void *node)
{
	std::vector<std::thread> th;
	th.reserve(n);   // (before the first thread exists: growing the vector must not be what throws while joinable threads sit in it)
	unsigned started = 1;
	try {
		for (; started < n; ++started) th.emplace_back([=] { stay_on_node(node); fn(arg, started); });
	} catch (const std::system_error&) {}
	fn(arg, 0);
	for (unsigned t = started; t < n; ++t) fn(arg, t);
	for (auto &x : th) x.join();
}
}   // namespace

// The CPUs this process may keep busy: its affinity mask, and the CPU-time quota of its control group where there is one (a
// container given "16 CPUs" of a 256-thread host sees all 256 and is stopped for the rest of every 100 ms period once its threads
// have used 1.6 CPU-seconds of it: 32 busy threads run for 50 ms and stand still for 50 -- measured on the MI355X boxes as phases
// of the 100 M-triangle walk that took twice as long in one pass as in the next).  HRY_CPUS overrides.
unsigned cpu_allowance()
{
	static const unsigned n = [] {
		if (const char *e = getenv("HRY_CPUS")) { int v = atoi(e); if (v > 0) return (unsigned)v; }
		cpu_set_t cs;
		CPU_ZERO(&cs);
		unsigned hw = sched_getaffinity(0, sizeof cs, &cs) == 0 ? (unsigned)CPU_COUNT(&cs) : std::thread::hardware_concurrency();
		if (!hw) hw = 1;
		double quota = 0;   // CPUs; 0 = none
		if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "max 100000" or "1600000 100000"
			char a[32] = {};
			double period = 0;
			if (fscanf(f, "%31s %lf", a, &period) == 2 && a[0] != 'm' && period > 0) quota = atof(a) / period;
			fclose(f);
		} else {   // cgroup v1
			double q = -1, period = 0;
			if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &q) != 1) q = -1; fclose(g); }
			if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &period) != 1) period = 0; fclose(g); }
			if (q > 0 && period > 0) quota = q / period;
		}
		if (quota >= 1.0) hw = std::min(hw, (unsigned)quota);
		else if (quota > 0) hw = 1;
		return hw;
	}();
	return n;
}

// fn(arg, t) for t = 0 .. n - 1, index 0 on the calling thread; returns when all have returned.  fn must not throw.
void run_on_helpers(unsigned n, void (*fn)(void*, unsigned), void *arg, const void *cpus)
{
	if (n <= 1) { fn(arg, 0); return; }
	const void *node = cpus ? cpus : callers_node_cpus();
	static AtFork at_fork;
	// a helper that starts a parallel phase of its own gets threads for that call only: a persistent set per helper would
	// grow to N x N parked threads
	if (t_is_helper) { run_on_fresh_threads(n, fn, arg, node); return; }
	Pool &P = t_pool;
	if (!P.st) P.st = std::make_shared<PoolState>();
	PoolState &S = *P.st;
	unsigned n_par = n;
	{
		std::unique_lock<std::mutex> lk(S.mu);
		if (S.busy) { lk.unlock(); run_on_fresh_threads(n, fn, arg, node); return; }
		// the counter names a helper only once its thread exists: a later call waits for exactly `pending` of them
		while (S.n_helpers + 1 < n) {
			try { std::thread(helper_main, P.st, S.n_helpers + 1, S.gen).detach(); } catch (const std::system_error&) { break; }
			++S.n_helpers;
		}
		n_par = std::min(n, S.n_helpers + 1);
		S.fn = fn; S.arg = arg; S.node = node; S.n_job = n_par; S.pending = n_par - 1; S.busy = true;
		++S.gen;
	}
	S.cv_work.notify_all();
	fn(arg, 0);
	for (unsigned t = n_par; t < n; ++t) fn(arg, t);   // indices whose helper could not be created
	std::unique_lock<std::mutex> lk(S.mu);
	S.cv_done.wait(lk, [&] { return S.pending == 0; });
	S.busy = false;
}

}   // namespace hry
