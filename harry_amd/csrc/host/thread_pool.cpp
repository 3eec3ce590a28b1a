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

#include <condition_variable>
#include <memory>
#include <mutex>
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

void helper_main(std::shared_ptr<PoolState> st, unsigned index, uint64_t seen)
{
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

void run_on_fresh_threads(unsigned n, void (*fn)(void*, unsigned), void *arg, const void *node)
{
	std::vector<std::thread> th;
	for (unsigned t = 1; t < n; ++t) th.emplace_back([=] { stay_on_node(node); fn(arg, t); });
	fn(arg, 0);
	for (auto &x : th) x.join();
}
}   // namespace

// fn(arg, t) for t = 0 .. n - 1, index 0 on the calling thread; returns when all have returned.  fn must not throw.
void run_on_helpers(unsigned n, void (*fn)(void*, unsigned), void *arg, const void *cpus)
{
	if (n <= 1) { fn(arg, 0); return; }
	const void *node = cpus ? cpus : callers_node_cpus();
	static AtFork at_fork;
	Pool &P = t_pool;
	if (!P.st) P.st = std::make_shared<PoolState>();
	PoolState &S = *P.st;
	{
		std::unique_lock<std::mutex> lk(S.mu);
		if (S.busy) { lk.unlock(); run_on_fresh_threads(n, fn, arg, node); return; }
		while (S.n_helpers + 1 < n) {
			const unsigned idx = ++S.n_helpers;
			std::thread(helper_main, P.st, idx, S.gen).detach();
		}
		S.fn = fn; S.arg = arg; S.node = node; S.n_job = n; S.pending = n - 1; S.busy = true;
		++S.gen;
	}
	S.cv_work.notify_all();
	fn(arg, 0);
	std::unique_lock<std::mutex> lk(S.mu);
	S.cv_done.wait(lk, [&] { return S.pending == 0; });
	S.busy = false;
}

}   // namespace hry
