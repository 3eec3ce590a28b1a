// Recycling pool for the big host arrays (mesh.hpp: BlockPool / BigVec).
#include <cstdio>
#include <cstdlib>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <new>
#include <unordered_map>

#include "mesh.hpp"

namespace hry {
namespace {
struct Pool {
	std::mutex mu;
	std::multimap<size_t, void*> free_blocks;        // capacity -> block
	std::unordered_map<void*, size_t> capacity;      // every block handed out by take(), live or free
	size_t free_bytes = 0, limit;
	Pool()
	{
		// HRY_POOL_MB: how much freed memory the pool may keep (0 disables recycling).  Default: an eighth of the machine's memory,
		// at least 2 GiB, at most 64 GiB -- a 100 M-triangle mesh frees 5 GB of per-call arrays per encode, and fresh ones cost
		// a page fault per 4 KiB (or per 2 MiB) under the process-wide mmap lock, which is what N concurrent workers then queue on
		const char *e = getenv("HRY_POOL_MB");
		size_t def = 2048;
		const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
		if (pages > 0 && psz > 0) def = std::min<size_t>(65536, std::max<size_t>(2048, ((size_t)pages * (size_t)psz >> 20) / 8));
		limit = (size_t)(e ? strtoull(e, nullptr, 10) : def) << 20;
	}
	~Pool() { for (auto &kv : free_blocks) free(kv.second); }
};
bool huge_pages_wanted() { static const bool on = getenv("HRY_NO_HUGEPAGES") == nullptr; return on; }
Pool &pool() { static Pool *p = new Pool(); return *p; }   // never destroyed: vectors in static objects may outlive any order
}   // namespace

void *BlockPool::take(size_t bytes)
{
	Pool &P = pool();
	{
		std::lock_guard<std::mutex> g(P.mu);
		auto it = P.free_blocks.lower_bound(bytes);
		if (it != P.free_blocks.end() && it->first <= bytes + bytes / 2 + (1u << 20)) {   // close enough in size
			void *p = it->second;
			P.free_bytes -= it->first;
			P.free_blocks.erase(it);
			return p;
		}
	}
	// 2 MiB alignment + MADV_HUGEPAGE: the walk and the replay touch these arrays ring by ring, i.e. a few entries on
	// thousands of different 4 KiB pages per ring -- with transparent huge pages (where the system grants them) the
	// page-table walks disappear.  Purely advisory: without THP nothing changes.
	const size_t huge = (size_t)2 << 20;
	const bool want_huge = bytes >= huge && huge_pages_wanted();
	const size_t cap = want_huge ? (bytes + huge - 1) & ~(huge - 1) : (bytes + 65535) & ~(size_t)65535;
	void *p = aligned_alloc(want_huge ? huge : 64, cap);
	if (!p) throw std::bad_alloc();
	if (want_huge) (void)madvise(p, cap, MADV_HUGEPAGE);
	std::lock_guard<std::mutex> g(P.mu);
	P.capacity[p] = cap;
	return p;
}

void BlockPool::give(void *p) noexcept
{
	if (!p) return;
	Pool &P = pool();
	std::lock_guard<std::mutex> g(P.mu);
	auto it = P.capacity.find(p);
	if (it == P.capacity.end()) { free(p); return; }
	const size_t cap = it->second;
	if (P.free_bytes + cap > P.limit) { P.capacity.erase(it); free(p); return; }
	P.free_blocks.emplace(cap, p);
	P.free_bytes += cap;
}

// ---- helper threads stay on the caller's memory node ------------------------------------------------------------
// A helper thread that first touches (or last wrote) a block leaves its pages on the node it ran on, and the recycling pool hands
// such blocks to the next call: on a two-socket host the sequential walk / replay of that call then works out of the other
// socket's memory (measured: replay 6.3 -> 35 ms per million triangles after a PLY parse whose twin matching ran on both
// sockets).  So every helper thread is confined to the CPUs of the node its creator is running on.
namespace {
struct NodeTable {
	std::vector<cpu_set_t> nodes;
	NodeTable()
	{
		if (getenv("HRY_NO_NUMA_BIND")) return;
		cpu_set_t allowed;
		CPU_ZERO(&allowed);
		if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
		for (int n = 0; n < 64; ++n) {
			char path[96];
			snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", n);
			FILE *f = fopen(path, "r");
			if (!f) break;
			cpu_set_t cs;
			CPU_ZERO(&cs);
			int a, b;
			while (fscanf(f, "%d", &a) == 1) {   // "0-63,128-191"
				b = a;
				int c = fgetc(f);
				if (c == '-') { if (fscanf(f, "%d", &b) != 1) break; c = fgetc(f); }
				for (int k = a; k <= b && k < CPU_SETSIZE; ++k) if (CPU_ISSET(k, &allowed)) CPU_SET(k, &cs);
				if (c != ',') break;
			}
			fclose(f);
			nodes.push_back(cs);
		}
		if (nodes.size() < 2) nodes.clear();   // one node: nothing to confine
	}
};
}   // namespace

// the CPUs that share the caller's last-level cache (one CCD of an EPYC: 8 cores).  What helper threads of a SHORT parallel phase
// write stays in their L3; a sequential phase that follows on the caller's core and rewrites those lines (the recycling pool hands
// the freed temporaries straight to the next call) pays a cache-to-cache transfer between chiplets per line -- as slow as remote
// memory (measured: the replay after a PLY parse 11 -> 23 ms even with every thread on the caller's memory node).
const void *callers_cache_cpus(unsigned *n_cpus)
{
	static std::mutex mu;
	static std::unordered_map<int, cpu_set_t> *by_cpu = new std::unordered_map<int, cpu_set_t>();
	if (n_cpus) *n_cpus = 0;
	if (getenv("HRY_NO_NUMA_BIND")) return nullptr;
	const int cpu = sched_getcpu();
	if (cpu < 0) return nullptr;
	std::lock_guard<std::mutex> g(mu);
	auto it = by_cpu->find(cpu);
	if (it == by_cpu->end()) {
		cpu_set_t cs, allowed;
		CPU_ZERO(&cs); CPU_ZERO(&allowed);
		(void)sched_getaffinity(0, sizeof(allowed), &allowed);
		char path[128];
		snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
		if (FILE *f = fopen(path, "r")) {
			int a, b;
			while (fscanf(f, "%d", &a) == 1) {
				b = a;
				int c = fgetc(f);
				if (c == '-') { if (fscanf(f, "%d", &b) != 1) break; c = fgetc(f); }
				for (int k = a; k <= b && k < CPU_SETSIZE; ++k) if (CPU_ISSET(k, &allowed)) CPU_SET(k, &cs);
				if (c != ',') break;
			}
			fclose(f);
		}
		it = by_cpu->emplace(cpu, cs).first;
	}
	const unsigned n = (unsigned)CPU_COUNT(&it->second);
	if (n < 2) return nullptr;
	if (n_cpus) *n_cpus = n;
	return &it->second;
}

const void *callers_node_cpus();
// the CPUs of the caller's last-level cache domain (else of its memory node) WITHOUT the caller's own core: for a helper that
// polls while the caller runs a sequential loop -- on the caller's sibling hardware thread it would take issue slots from it
const void *callers_neighbour_cpus()
{
	static std::mutex mu;
	static std::unordered_map<int, cpu_set_t> *by_cpu = new std::unordered_map<int, cpu_set_t>();
	if (getenv("HRY_NO_NUMA_BIND")) return nullptr;
	const int cpu = sched_getcpu();
	if (cpu < 0) return nullptr;
	const void *base = callers_cache_cpus(nullptr);
	if (!base) base = callers_node_cpus();
	std::lock_guard<std::mutex> g(mu);
	auto it = by_cpu->find(cpu);
	if (it == by_cpu->end()) {
		cpu_set_t cs;
		if (base) cs = *(const cpu_set_t*)base;
		else { CPU_ZERO(&cs); (void)sched_getaffinity(0, sizeof(cs), &cs); }
		CPU_CLR(cpu, &cs);
		char path[128];
		snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
		if (FILE *f = fopen(path, "r")) {
			int a, b;
			while (fscanf(f, "%d", &a) == 1) {
				b = a;
				int c = fgetc(f);
				if (c == '-') { if (fscanf(f, "%d", &b) != 1) break; c = fgetc(f); }
				for (int k = a; k <= b && k < CPU_SETSIZE; ++k) CPU_CLR(k, &cs);
				if (c != ',') break;
			}
			fclose(f);
		}
		it = by_cpu->emplace(cpu, cs).first;
	}
	return CPU_COUNT(&it->second) > 0 ? &it->second : nullptr;
}

static const NodeTable &node_table() { static const NodeTable *tab = new NodeTable(); return *tab; }
const void *node_cpus(int node)
{
	const NodeTable &tab = node_table();
	if (node < 0 || (size_t)node >= tab.nodes.size() || CPU_COUNT(&tab.nodes[node]) == 0) return nullptr;
	return &tab.nodes[node];
}
const void *callers_node_cpus()
{
	const NodeTable *tab = &node_table();
	if (tab->nodes.empty()) return nullptr;
	const int cpu = sched_getcpu();
	if (cpu < 0) return nullptr;
	for (const cpu_set_t &cs : tab->nodes) if (CPU_ISSET(cpu, &cs) && CPU_COUNT(&cs) > 0) return &cs;
	return nullptr;
}
void stay_on_node(const void *cpus)
{
	if (cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), (const cpu_set_t*)cpus);
}

}   // namespace hry
