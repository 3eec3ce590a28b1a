// Host-only pieces of the product under AddressSanitizer / UBSan (CPU build, no HIP): PLY and OBJ readers and writers, twin
// matching, the cut-border walk, the reference-stream reader with its replay, header parsing of arbitrary bytes, sharding.
// Built and run by tests/test_host_cpu.py::test_host_code_under_sanitizers.  Usage: driver FILE...  (.ply .obj .hry)
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include "../../harry_amd/csrc/host/host.hpp"
#include "../../harry_amd/csrc/host/cbm_replay.hpp"
#include <atomic>
#include <thread>

using namespace hry;

static std::vector<uint8_t> slurp(const std::string &fn)
{
	std::ifstream is(fn, std::ios::binary);
	return std::vector<uint8_t>((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv)
{
	int done = 0;
	for (int i = 1; i < argc; ++i) {
		const std::string fn = argv[i];
		const std::vector<uint8_t> data = slurp(fn);
		const std::string ext = fn.size() >= 4 ? fn.substr(fn.size() - 4) : "";
		try {
			if (ext == ".ply" || ext == ".obj") {
				const std::string dir = fn.substr(0, fn.find_last_of('/'));
				std::unique_ptr<Mesh> m(ext == ".ply" ? mesh_from_ply(data.data(), data.size()) : mesh_from_obj(data.data(), data.size(), dir.c_str()));
				ensure_twins(*m);
				ByteSink out;
				mesh_to_ply(*m, true, out);
				mesh_to_ply(*m, false, out, true);
				mesh_to_obj(*m, out);
				for (auto &L : m->lists) { L.bmin.assign(L.stride(), 0); L.bmax.assign(L.stride(), 0); L.have_bounds = true; }
				std::vector<uint8_t> hdr;
				write_hry_header(*m, 1, hdr);
				Mesh back;
				int minor = 0;
				read_hry_header(hdr.data(), hdr.size(), back, minor);
				if (m->nf) {
					ShardPlan plan;
					shard_plan(*m, 3, plan);
					// a sharded container as three ranks would write it (host side: header of the whole mesh, run tables; the bodies
					// are stand-ins), merged, then its directory under the checks every reader makes -- intact and damaged
					std::vector<std::vector<uint8_t>> parts;
					for (uint32_t s = 0; s < 3; ++s) {
						std::unique_ptr<Mesh> sh(shard_extract(*m, plan, s));
						std::vector<uint8_t> c;
						write_hry_header(*sh, 3, c);
						auto put = [&](const void *p, size_t n) { c.insert(c.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
						const uint32_t one = sh->nf ? 1 : 0, nr = (uint32_t)sh->shard.runs.size();
						put(&one, 4);
						if (one) {
							const size_t nl2 = sh->general ? 2 * sh->lists.size() : 0;   // general bindings: every run with its record ranges
							const uint64_t len = 4 + (sizeof(ShardRun) + 4 * nl2) * (uint64_t)nr + 7;
							put(&len, 8); put(&nr, 4);
							for (uint32_t j = 0; j < nr; ++j) { put(&sh->shard.runs[j], sizeof(ShardRun)); if (nl2) put(sh->shard.run_records.data() + j * nl2, 4 * nl2); }
							put("body...", 7);
						}
						parts.push_back(std::move(c));
					}
					std::vector<const uint8_t*> pp; std::vector<size_t> ps;
					for (auto &c : parts) { pp.push_back(c.data()); ps.push_back(c.size()); }
					ByteSink merged;
					merge_containers(pp.data(), ps.data(), pp.size(), merged);
					Mesh hm; int mn3 = 0;
					const size_t h3 = read_hry_header(merged.data(), merged.size(), hm, mn3, false);
					ShardedDirectory dir;
					std::vector<uint32_t> counts;
					for (const AttrList &L : hm.lists) counts.push_back(L.count);
					const std::vector<uint32_t> *lc = hm.general ? &counts : nullptr;
					parse_sharded_directory(merged.data(), merged.size(), h3, hm.nv, hm.nf, hm.declared_ne, dir, false, lc);
					if (!dir.complete || mn3 != 3) throw Error(HRY_E_INTERNAL, "merged directory incomplete");
					for (int k = 0; k < 64; ++k) {
						std::vector<uint8_t> bad(merged.begin(), merged.end());
						if (k < 8) bad.resize(h3 + (size_t)k * (bad.size() - h3) / 8);
						else for (int j = 0; j < 2; ++j) bad[h3 + (size_t)(1103515245u * (unsigned)(k * 2 + j + 1) + 12345u) % (bad.size() - h3)] ^= (uint8_t)(1u << ((k + j) & 7));
						try {
							ShardedDirectory d2;
							parse_sharded_directory(bad.data(), bad.size(), h3, hm.nv, hm.nf, hm.declared_ne, d2, (k & 1) != 0, lc);
							const uint8_t *bp = bad.data(); const size_t bn = bad.size();
							ByteSink again;
							merge_containers(&bp, &bn, 1, again);
						} catch (const Error &) {
						}
					}
				}
				if (m->nf) {
					WalkResult w; cut_border_walk(*m, w);
					if (m->general) {   // which record every element names: with the symbols' positions, and without them
						Events E1, E2;
						collect_events(*m, w, w.n_conn, true, E1);
						collect_events(*m, w, 0, false, E2);
						for (size_t l = 0; l < m->lists.size(); ++l)
							if (E1.ls[l].created != E2.ls[l].created || E1.ls[l].type_sym.size() != E2.ls[l].type_sym.size()) throw Error(HRY_E_INTERNAL, "events: the two collections differ");
					}
				}
				if (m->nf && !m->general) {
					// the walk of the chunked profile (no operation model): on the host threads where the mesh has several components
					// (the test sets HRY_PARALLEL_MIN_FACES=1), its components found by the threads' union-find, coded where they belong
					std::unique_ptr<Mesh> m2(ext == ".ply" ? mesh_from_ply(data.data(), data.size()) : mesh_from_obj(data.data(), data.size(), dir.c_str()));
					ensure_twins(*m2);
					WalkResult w2;
					w2.snapshot_faces = 50;   // (border snapshots inside the components: restart points for the replay below)
					cut_border_walk(*m2, w2, false);
					if (w2.order_f.size() != m2->nf) throw Error(HRY_E_INTERNAL, "plain walk: a face was not coded");
					{
						// the decoder's replay of what the walk wrote: as one sequence, and from the restart points and border snapshots of the
						// directory on the host threads (spans joined afterwards) -- the same arrays; then the snapshots' section damaged
						std::vector<uint8_t> planes[21];
						static const int first_plane[G_COUNT] = { 0, 1, 5, 7, 11 };
						for (int g = 0; g < G_COUNT; ++g)
							for (int b = 0; b < kGroupBytes[g]; ++b) {
								std::vector<uint8_t> &pl = planes[first_plane[g] + b];
								pl.resize(w2.grp_val[g].size());
								for (size_t q = 0; q < pl.size(); ++q) pl[q] = (uint8_t)(w2.grp_val[g][q] >> (8 * b));
							}
						for (size_t q = 0; q < w2.op_sc.size(); ++q) planes[13 + (op_u8(w2.op_sc[q]) >> 3)].push_back(op_u8(w2.op_sc[q]) & 7);
						PlaneView views[21];
						for (int k = 0; k < 21; ++k) views[k] = PlaneView(planes[k]);
						auto skeleton = [&](Mesh &d) { d.nv = m2->nv; d.nf = m2->nf; d.declared_ne = m2->ne(); d.have_degree = m2->have_degree; };
						Mesh seq, par;
						skeleton(seq); skeleton(par);
						OrderVec ov_seq, ov_par;
						std::vector<uint32_t> ss0, sl0, ss1, sl1;
						const std::vector<RestartPoint> no_points;
						const std::vector<RestartCounters> no_counters;
						cut_border_replay(seq, views, no_points, no_counters, ov_seq, ss0, sl0);
						std::vector<RestartCounters> rc, sc;
						const std::vector<RestartPoint> rs = select_restart_points(w2.marks, w2.named, rc, &w2.snapshots, &sc);
						std::vector<uint8_t> sec;
						std::vector<SnapshotPoint> snaps;
						if (!w2.snapshots.empty()) {
							write_snapshot_section(w2.snapshot_faces, w2.snapshots, sc, sec);
							uint32_t spacing = 0;
							if (read_snapshot_section(sec.data(), sec.size(), m2->nv, spacing, snaps) != sec.size()) throw Error(HRY_E_INTERNAL, "snapshots: the section does not read back");
						}
						cut_border_replay(par, views, rs, rc, ov_par, ss1, sl1, nullptr, &snaps);
						if (seq.org != par.org || seq.twin != par.twin || seq.face_off != par.face_off || ov_seq != ov_par || ss0 != ss1 || sl0 != sl1) throw Error(HRY_E_INTERNAL, "replay from the directory's points differs from the sequential replay");
						// ... and the way the pipelined decode of a triangle mesh takes (device/unchunk.cpp): the calling thread replays the
						// stretch up to the first snapshot and publishes its progress, helper threads the stretches behind the snapshots
						// (SnapshotSpans), joined one after the other as they finish; a reader thread plays the consumer
						int udeg = 0;
						if (!snaps.empty() && rs.empty() && planes[7].empty() && m2->uniform_degree(udeg) && udeg == 3) {
							Mesh live_m;
							skeleton(live_m);
							live_m.face_off.resize((size_t)live_m.nf + 1); live_m.face_off[0] = 0;
							live_m.org.resize(live_m.declared_ne);
							OrderVec ov_live;
							ov_live.assign(live_m.nv, 0);
							SnapshotSpans spans(live_m, views, snaps, ov_live.data());   // (sizes the twins)
							BigVec<uint16_t> seen(live_m.nv, 0);
							ReplayLive live;
							live.on_border.assign(live_m.nv, 0);
							live.interval = 64;
							std::atomic<bool> stop{ false };
							uint64_t n_ranges = 0, n_pubs = 0;
							std::thread consumer([&] {
								uint64_t seen_seq = 0;
								for (;;) {
									while (live.announced.load(std::memory_order_acquire) == seen_seq && !stop.load()) std::this_thread::yield();
									std::unique_lock<std::mutex> lk(live.mu);
									const ReplayLive::Pub P = live.pub;
									n_ranges += live.ranges.size(); live.ranges.clear();
									live.patches.clear();
									seen_seq = P.seq;
									++n_pubs;
									if (P.done || P.failed || stop.load()) break;
								}
							});
							ReplayCursor cur;
							std::vector<uint32_t> cf;
							std::vector<std::pair<uint32_t, uint32_t>> refs;
							try {
								spans.announce_to = &live;
								spans.start(3);
								BorderEnd end0;
								size_t cur_end0[21];
								const bool eom0 = replay_triangles<true>(live_m, views, seen.data(), ov_live.data(), cur, cf, refs, &live, nullptr, spans.spans[0].cur1, spans.spans[0].stop_face, true, nullptr, &end0, cur_end0);
								if (!eom0) live.publish(cur.face, cur.he, cur.next_id, false);
								spans.finish(cur, cur_end0, std::move(end0), eom0, &live);
								live.publish(cur.face, cur.he, cur.next_id, true);
							} catch (...) { stop.store(true); live.publish(cur.face, cur.he, cur.next_id, true, true); consumer.join(); throw; }
							consumer.join();
							ov_live.resize(cur.next_id);
							if (live_m.org != seq.org || live_m.twin != seq.twin || live_m.face_off != seq.face_off || ov_live != ov_seq || n_ranges != snaps.size())
								throw Error(HRY_E_INTERNAL, "publishing replay with stretches on helper threads differs from the sequential replay");
						}
#if !defined(__SANITIZE_THREAD__)   // (spans started from a damaged directory may overlap: every index is checked against the header's sizes, the result is an error -- but two threads may have written the same word)
						for (size_t k = 0; k < sec.size() && k < 4000; ++k) {
							std::vector<uint8_t> bad = sec;
							bad[k] ^= (uint8_t)(1u << (k % 8));
							try {
								std::vector<SnapshotPoint> sn2;
								uint32_t spacing = 0;
								read_snapshot_section(bad.data(), bad.size(), m2->nv, spacing, sn2);
								Mesh d;
								skeleton(d);
								OrderVec ov;
								std::vector<uint32_t> a, b;
								cut_border_replay(d, views, rs, rc, ov, a, b, nullptr, &sn2);
							} catch (const Error &) {
							}
						}
#endif
					}
					// ... and a shard of it walked in place on the whole mesh's arrays (the in-process executor's walk)
					std::unique_ptr<Mesh> m3(ext == ".ply" ? mesh_from_ply(data.data(), data.size()) : mesh_from_obj(data.data(), data.size(), dir.c_str()));
					ensure_twins(*m3);
					ShardPlan light;
					shard_plan(*m3, 2, light, true);
					int ud = 0;
					const bool uniform = m3->uniform_degree(ud) && (ud == 3 || ud == 4);
					BigVec<uint32_t> eface;
					if (!uniform && light.A.eface.size() != m3->ne()) {
						eface.resize(m3->ne());
						for (uint32_t f = 0; f < m3->nf; ++f) for (uint32_t h = m3->face_off[f]; h < m3->face_off[f + 1]; ++h) eface[h] = f;
					}
					WalkState marks(m3->nv, m3->nf);
					for (uint32_t sidx = 0; sidx < 2; ++sidx) {
						ComponentAnalysis part;
						ShardInfo info;
						shard_components(light, sidx, part, info);
						if (!part.ncomp) continue;
						WalkResult w3;
						cut_border_walk_in_place(*m3, part, uniform ? nullptr : eface.empty() ? light.A.eface.data() : eface.data(), marks, w3);
					}
				}
				// the same text, damaged: a clean error or a mesh
				for (int k = 0; k < 16; ++k) {
					std::vector<uint8_t> bad = data;
					for (int j = 0; j < 4; ++j) bad[(size_t)(1103515245u * (unsigned)(k * 4 + j + 1) + 12345u) % bad.size()] = (uint8_t)("0 /-\n9fv"[(k + j) % 9]);
					try {
						std::unique_ptr<Mesh> b(ext == ".ply" ? mesh_from_ply(bad.data(), bad.size()) : mesh_from_obj(bad.data(), bad.size(), dir.c_str()));
						ensure_twins(*b);
					} catch (const Error &) {
					} catch (const std::bad_alloc &) {
					} catch (const std::length_error &) {
					}
				}
			} else if (ext == ".hry") {
				Mesh m;
				int minor = 0;
				const size_t h = read_hry_header(data.data(), data.size(), m, minor);
				if (minor == 1) {
					OrderVec order_v;
					std::vector<uint32_t> seg_start, seg_level;
					if (m.general) {
						std::vector<GenRecordEvents> ev;
						std::vector<uint8_t> planes;
						read_general_stream(data.data() + h, data.size() - h, m, order_v, ev, seg_start, seg_level, -1, planes);
					} else {
						std::vector<uint8_t> vp, fp;
						read_compat_stream(data.data() + h, data.size() - h, m, order_v, seg_start, seg_level, vp, fp);
					}
				}
				// the same bytes, damaged: must fail cleanly or decode something
				for (int k = 0; k < 24; ++k) {
					if (getenv("HRY_DRIVER_VERBOSE")) fprintf(stderr, "k=%d t=%ld\n", k, (long)clock() / 1000);
					std::vector<uint8_t> bad = data;
					for (int j = 0; j < 3; ++j) bad[(size_t)(1103515245u * (unsigned)(k * 3 + j + 1) + 12345u) % bad.size()] ^= (uint8_t)(1 + k + j);
					try {
						int mn = 0;
						{   // a damaged record count is a legal header naming gigabytes: nothing to learn from filling them
							Mesh probe;
							read_hry_header(bad.data(), bad.size(), probe, mn, false);
							uint64_t bytes = ((uint64_t)probe.nv + probe.nf) * 16;
							for (auto &L : probe.lists) bytes += (uint64_t)L.count * (L.stride() + 1);
							if (bytes > (64u << 20)) continue;
						}
						Mesh b;
						const size_t hb = read_hry_header(bad.data(), bad.size(), b, mn);
						if (getenv("HRY_DRIVER_VERBOSE")) fprintf(stderr, "  minor %d nv %u nf %u general %d\n", mn, b.nv, b.nf, (int)b.general);
						if (mn == 1 && (uint64_t)b.nv + b.nf < (1u << 22)) {
							OrderVec ov;
							std::vector<uint32_t> ss, sl;
							if (b.general) { std::vector<GenRecordEvents> ev; std::vector<uint8_t> pl; read_general_stream(bad.data() + hb, bad.size() - hb, b, ov, ev, ss, sl, -1, pl); }
							else { std::vector<uint8_t> vp, fp; read_compat_stream(bad.data() + hb, bad.size() - hb, b, ov, ss, sl, vp, fp); }
						}
					} catch (const Error &) {
					} catch (const std::bad_alloc &) {
					} catch (const std::length_error &) {
					}
				}
			}
			++done;
		} catch (const Error &e) {
			fprintf(stderr, "%s: %s\n", fn.c_str(), e.what());
			return 2;
		}
	}
	printf("ok %d files\n", done);
	return 0;
}
