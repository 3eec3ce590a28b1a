// Development: what one parallel_for costs in thread creation + joins (g++ -O2 -pthread scripts/spawn_cost.cpp -o /tmp/spawn_cost)
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <thread>
#include <vector>
#include <sched.h>
int main(int argc, char **argv)
{
	int n = argc > 1 ? atoi(argv[1]) : 32;
	cpu_set_t cs;
	sched_getaffinity(0, sizeof cs, &cs);
	for (int rep = 0; rep < 5; ++rep) {
		auto t0 = std::chrono::steady_clock::now();
		for (int k = 0; k < 20; ++k) {
			std::vector<std::thread> th;
			for (int t = 1; t < n; ++t) th.emplace_back([&] { sched_setaffinity(0, sizeof cs, &cs); });
			for (auto &x : th) x.join();
		}
		double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		printf("%d threads: %.3f ms per spawn + join of all\n", n, ms / 20);
	}
}
