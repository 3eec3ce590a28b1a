// Recycling pool for the big host arrays (mesh.hpp: BlockPool / BigVec).
#include <cstdlib>
#include <sys/mman.h>
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
		// HRY_POOL_MB: how much freed memory the pool may keep (default 2048 MiB, 0 disables recycling)
		const char *e = getenv("HRY_POOL_MB");
		limit = (size_t)(e ? strtoull(e, nullptr, 10) : 2048ull) << 20;
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

}   // namespace hry
