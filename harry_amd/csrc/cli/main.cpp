// harry -- the reference's command line (main.cc:23-123) on top of the C ABI of libharry_amd.so (include/harry_amd.h):
//     harry [OPTIONS] INPUT OUTPUT      -f/--format hry|ply|obj   -l/--list L   -a/--attr A   -q/--quant Q   -c/--clear-quant
//                                       --ply-ascii   -h/--help
// Same flag state machine (-l sets the list, -a the component for the next -q only, main.cc:47-71), same phase prints
// (main.cc:99-120), same error texts; an error ends the program the way the reference's uncaught std::runtime_error does
// (message of std::terminate on stderr, status 134).  Input kind by magic number, output kind by extension
// (formats/unified_reader.h:33-56, unified_writer.h:31-47): .hry, .ply and .obj both ways.
// Additive options (the reference rejects them as invalid): --profile compat|chunked (default compat = the reference's own
// v0.1 stream), --chunk N, --device D, --gpus N (N device contexts behind this one command, devices D, D+1, ... -- more
// contexts than devices share them: the mesh shards by connected component, every context codes / decodes its shards on a
// worker thread of its own, one sharded container comes out; hry_encode_sharded / hry_decode_sharded), --shards N (number of
// shards, default one per context; both imply --profile chunked, the reference's single stream does not shard), --ply-packed
// (binary PLY of a quantised mesh with every value in the width its header declares; the reference's writer dumps the
// original-width records, formats/ply/writer.cc:72-75).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/harry_amd.h"

namespace {

const std::chrono::steady_clock::time_point g_start = std::chrono::steady_clock::now();   // (static initialisation: as early as this program can look)
long long since_start_ms() { return (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - g_start).count(); }
struct QuantArg { int l, o, q; };
struct Args {
	std::string in, out, fmt, profile;   // profile empty: compat, unless --gpus / --shards ask for the sharded container
	std::vector<QuantArg> quant;
	bool clearquant = false, ply_ascii = false, ply_packed = false;
	int chunk = 0, device = 0, shards = 0, gpus = 0;
};

struct Opt { char s; const char *l; const char *descr; bool has_val; };
const Opt kOpts[] = {
	{ 'h', "help", "Print this dialogue.", false }, { 'f', "format", "Enforce output format", true }, { 'l', "list", "Select attribute list", true },
	{ 'a', "attr", "Select attribute", true }, { 'q', "quant", "Quantization bits", true }, { 'c', "clear-quant", "Clear all quantization first", false },
	{ 0, "ply-ascii", "PLY writer: Use ASCII format", false }, { 0, "ply-packed", "PLY writer: quantised values in their declared width", false }, { 0, "profile", "compat (reference stream, default) or chunked", true },
	{ 0, "chunk", "chunked: symbols per chunk", true }, { 0, "device", "GPU index (first one with --gpus)", true }, { 0, "gpus", "N device contexts, one worker thread each", true },
	{ 0, "shards", "chunked: code as N shards (default: one per context)", true } };

void usage(const char *argv0)
{
	std::cout << "Harry mesh compressor" << std::endl << std::endl << "Usage: " << argv0 << " [OPTIONS] INPUT OUTPUT" << std::endl;
	for (const Opt &o : kOpts) {
		std::cout << "  ";
		if (o.s) std::cout << '-' << o.s << ", "; else std::cout << "    ";
		char buf[64];
		snprintf(buf, sizeof buf, "--%-13s", o.l);
		std::cout << buf << o.descr << std::endl;
	}
}
[[noreturn]] void arg_error(const char *argv0, const std::string &what)
{
	usage(argv0);
	std::cout << std::endl;
	std::cerr << "Error: " << what << std::endl;
	std::exit(EXIT_FAILURE);
}
int to_int(const char *argv0, const std::string &v)
{
	char *end = nullptr;
	long x = strtol(v.c_str(), &end, 10);
	if (v.empty() || *end) arg_error(argv0, "Invalid cast");
	return (int)x;
}

Args parse(int argc, const char **argv)
{
	Args a;
	int cur_l = -1, cur_a = -1;   // (the reference leaves cur_l uninitialised, main.cc:48: -q without -l is reported instead)
	bool have_l = false;
	std::vector<std::string> pos;
	for (int i = 1; i < argc; ++i) {
		std::string s = argv[i];
		const Opt *o = nullptr;
		std::string val;
		bool inline_val = false;
		if (s.size() > 2 && s[0] == '-' && s[1] == '-') {
			std::string name = s.substr(2);
			size_t eq = name.find('=');
			if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); inline_val = true; }
			for (const Opt &k : kOpts) if (name == k.l) o = &k;
			if (!o) arg_error(argv[0], "Invalid option " + name);
		} else if (s.size() >= 2 && s[0] == '-' && s[1] != '-') {
			for (const Opt &k : kOpts) if (k.s && k.s == s[1]) o = &k;
			if (!o) arg_error(argv[0], std::string("Invalid option ") + s[1]);
			if (s.size() > 2) { val = s.substr(2); inline_val = true; }
		} else { pos.push_back(s); continue; }
		if (o->has_val && !inline_val) {
			if (i + 1 >= argc) arg_error(argv[0], "Too few non-optional arguments");
			val = argv[++i];
		}
		const std::string n = o->l;
		if (n == "help") { usage(argv[0]); std::exit(EXIT_SUCCESS); }
		else if (n == "format") { if (val != "hry" && val != "ply" && val != "obj") arg_error(argv[0], "Invalid enum value"); a.fmt = val; }
		else if (n == "list") { cur_l = to_int(argv[0], val); have_l = true; }
		else if (n == "attr") cur_a = to_int(argv[0], val);
		else if (n == "quant") {
			if (!have_l) arg_error(argv[0], "-q needs a preceding -l (the reference reads an uninitialised list index here)");
			a.quant.push_back(QuantArg{ cur_l, cur_a, to_int(argv[0], val) });
			cur_a = -1;
		}
		else if (n == "clear-quant") a.clearquant = true;
		else if (n == "ply-ascii") a.ply_ascii = true;
		else if (n == "ply-packed") a.ply_packed = true;
		else if (n == "profile") { if (val != "compat" && val != "chunked") arg_error(argv[0], "Invalid enum value"); a.profile = val; }
		else if (n == "chunk") a.chunk = to_int(argv[0], val);
		else if (n == "device") a.device = to_int(argv[0], val);
		else if (n == "shards") a.shards = to_int(argv[0], val);
		else if (n == "gpus") a.gpus = to_int(argv[0], val);
	}
	if (a.gpus < 0 || a.shards < 0 || a.gpus > 64) arg_error(argv[0], "Invalid number of contexts / shards");
	if ((a.gpus > 1 || a.shards > 1) && a.profile == "compat") arg_error(argv[0], "--gpus / --shards need --profile chunked: the reference's single stream does not shard");
	if (a.profile.empty()) a.profile = a.gpus > 1 || a.shards > 1 ? "chunked" : "compat";
	if (pos.size() < 2) arg_error(argv[0], "Too few non-optional arguments");
	if (pos.size() > 2) arg_error(argv[0], "Too much non-optional arguments");
	a.in = pos[0]; a.out = pos[1];
	return a;
}

void ok(int rc) { if (rc != HRY_OK) throw std::runtime_error(hry_last_error()); }

std::string ext_of(const std::string &fn)
{
	std::string e = fn.size() >= 4 ? fn.substr(fn.size() - 4) : std::string();
	std::transform(e.begin(), e.end(), e.begin(), ::tolower);
	return e;
}

// the input file, mapped read-only (an empty file, or one that cannot be mapped, is read the plain way)
struct MappedFile {
	const uint8_t *p = nullptr;
	size_t n = 0;
	bool mapped = false;
	std::vector<uint8_t> copy;
	explicit MappedFile(const std::string &path)
	{
		const int fd = open(path.c_str(), O_RDONLY);
		if (fd < 0) throw std::runtime_error("cannot open " + path);
		struct stat sb;
		if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
			void *q = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
			if (q != MAP_FAILED) { p = (const uint8_t*)q; n = (size_t)sb.st_size; mapped = true; (void)madvise(q, n, MADV_WILLNEED); }
		}
		if (!mapped) {
			uint8_t buf[1 << 16];
			for (;;) { const ssize_t k = read(fd, buf, sizeof buf); if (k <= 0) break; copy.insert(copy.end(), buf, buf + k); }
			p = copy.data(); n = copy.size();
		}
		close(fd);
	}
	~MappedFile() { if (mapped) munmap((void*)p, n); }
	MappedFile(const MappedFile&) = delete;
	size_t size() const { return n; }
	const uint8_t *data() const { return p; }
	uint8_t operator[](size_t i) const { return p[i]; }
};

// the output file; a large one is written by a few threads, each its own range (the copy into the page cache is the work:
// 1.7 GB of PLY took 0.4 s on one thread)
void write_file(const std::string &path, const uint8_t *p, size_t n)
{
	const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
	if (fd < 0) throw std::runtime_error("cannot write " + path);
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	const unsigned nt = n >= ((size_t)64 << 20) ? std::min(8u, hw) : 1u;
	std::vector<std::thread> th;
	std::vector<char> failed(nt, 0);
	auto part = [&](unsigned t) {
		size_t at = n * t / nt;
		const size_t end = n * (t + 1) / nt;
		while (at < end) {
			const ssize_t k = pwrite(fd, p + at, std::min<size_t>(end - at, (size_t)64 << 20), (off_t)at);
			if (k <= 0) { failed[t] = 1; return; }
			at += (size_t)k;
		}
	};
	for (unsigned t = 1; t < nt; ++t) th.emplace_back(part, t);
	part(0);
	for (auto &x : th) x.join();
	const bool bad = std::find(failed.begin(), failed.end(), (char)1) != failed.end();
	if (close(fd) != 0 || bad) throw std::runtime_error("cannot write " + path);
}

struct Handles {   // released on every path
	std::vector<hry_ctx*> cx;   // [0]: the context of every single-device step
	hry_mesh *mesh = nullptr;
	std::vector<uint8_t*> bufs;
	~Handles()
	{
		for (uint8_t *b : bufs) hry_free(b);
		if (mesh) hry_mesh_free(mesh);
		for (hry_ctx *c : cx) hry_ctx_destroy(c);
	}
};

int run(const Args &args)
{
	typedef std::chrono::high_resolution_clock Clock;
	auto ms = [](Clock::time_point a, Clock::time_point b) { return (long long)std::chrono::duration_cast<std::chrono::milliseconds>(b - a).count(); };
	Handles h;
	const int n_ctx = std::max(1, args.gpus);
	const int n_dev = hry_device_count();
	// The device contexts come up (runtime start, code objects: a few hundred milliseconds) while the input is read and parsed on the
	// host; whoever needs one first waits for them.
	std::string ctx_error;
	std::thread ctx_thread([&] {
		for (int i = 0; i < n_ctx; ++i) {
			hry_ctx *c = nullptr;
			if (hry_ctx_create(n_dev > 0 && i > 0 ? (args.device + i) % n_dev : args.device, &c) != HRY_OK) { ctx_error = hry_last_error(); return; }
			h.cx.push_back(c);
		}
	});
	struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{ ctx_thread };
	const bool trace = getenv("HRY_TRACE") != nullptr;
	auto contexts = [&] {
		if (ctx_thread.joinable()) { ctx_thread.join(); if (trace) std::cerr << "[harry] " << since_start_ms() << " ms  contexts ready" << std::endl; }
		if (!ctx_error.empty()) throw std::runtime_error(ctx_error);
	};
	if (trace) std::cerr << "[harry] " << since_start_ms() << " ms  arguments parsed" << std::endl;
	const bool sharded = n_ctx > 1 || args.shards > 1;

	std::cout << "Reading input..." << std::endl;
	Clock::time_point t0 = Clock::now();
	// the file is mapped, not copied: the parser's threads fault its pages in as they go (a 1.6 GB PLY read into a zero-filled
	// vector first was 0.6 s before the first byte was looked at)
	MappedFile in(args.in);
	if (in.size() >= 4 && in[0] == 0xfa && in[1] == 0xff && in[2] == 0xaf && in[3] == 0xaf) {
		contexts();
		if (n_ctx > 1) {
			hry_shard_timing st{};
			ok(hry_decode_sharded(h.cx.data(), n_ctx, in.data(), in.size(), nullptr, &h.mesh, &st));
			std::cout << "  " << st.n_segments << " segment(s) on " << st.n_contexts << " context(s), " << std::min(n_ctx, std::max(1, n_dev)) << " device(s): decode "
			          << (long long)st.encode_ms << " ms, placement " << (long long)st.extract_ms << " ms" << std::endl;
		} else ok(hry_decode(h.cx[0], in.data(), in.size(), nullptr, &h.mesh));
	}
	else if (in.size() >= 3 && in[0] == 'p' && in[1] == 'l' && in[2] == 'y') ok(hry_mesh_from_ply(in.data(), in.size(), &h.mesh));
	else if (ext_of(args.in) == ".obj") {
		// material libraries are looked up in the input path up to its last separator -- the whole path when there is none, as
		// the reference computes it (formats/unified_reader.h:56)
		const std::string dir = args.in.substr(0, args.in.find_last_of("/\\"));
		ok(hry_mesh_from_obj(in.data(), in.size(), dir.c_str(), &h.mesh));
		std::cout << "Used face regions: " << hry_mesh_nregions(h.mesh, 0) << std::endl;     // formats/obj/reader.rl:296-297
		std::cout << "Used vertex regions: " << hry_mesh_nregions(h.mesh, 1) << std::endl;
	}
	else throw std::runtime_error("Not a mesh file");
	if (trace) std::cerr << "[harry] " << since_start_ms() << " ms  input parsed" << std::endl;
	contexts();
	Clock::time_point t1 = Clock::now();
	std::cout << "Reading input took " << ms(t0, t1) << " ms." << std::endl;

	std::string type = args.fmt;
	if (type.empty()) {
		const std::string e = ext_of(args.out);
		type = e == ".hry" ? "hry" : e == ".ply" ? "ply" : e == ".obj" ? "obj" : "";
	}
	std::vector<hry_quant> q;
	for (const QuantArg &x : args.quant) q.push_back(hry_quant{ x.l, x.o, x.q });
	// a sharded .hry quantises shard by shard on the workers (the bounds of the whole mesh are combined from the shards'): the
	// whole mesh never sits on one device
	const bool quant_in_encode = sharded && type == "hry";
	const bool quant_phase = (!q.empty() || args.clearquant) && !quant_in_encode;
	if (quant_phase) {
		std::cout << "Quantization..." << std::endl;
		ok(hry_requant(h.cx[0], h.mesh, q.data(), q.size(), args.clearquant ? 1 : 0));   // validation and texts of main.cc:74-91 inside
	}
	Clock::time_point t2 = Clock::now();
	if (quant_phase) std::cout << "Quantization took " << ms(t1, t2) << " ms." << std::endl;

	std::cout << "Writing output..." << std::endl;
	if (type.empty()) throw std::runtime_error("Unknown file extension");
	uint8_t *out = nullptr;
	size_t out_len = 0;
	if (type == "hry") {
		hry_opts o{};
		o.profile = args.profile == "chunked" ? HRY_PROFILE_CHUNKED : HRY_PROFILE_COMPAT;
		o.chunk_syms = args.chunk;
		if (quant_in_encode) {
			if (o.profile != HRY_PROFILE_CHUNKED) throw std::runtime_error("--gpus / --shards need --profile chunked");
			o.shard_count = args.shards;
			hry_shard_timing st{};
			ok(hry_encode_sharded(h.cx.data(), n_ctx, h.mesh, q.data(), q.size(), args.clearquant ? 1 : 0, &o, &out, &out_len, &st));
			std::cout << "  " << st.n_shards << " shard(s) of " << st.n_components << " component(s) on " << st.n_contexts << " context(s), " << std::min(n_ctx, std::max(1, n_dev))
			          << " device(s): plan " << (long long)st.plan_ms << " ms, extract " << (long long)st.extract_ms << " ms, bounds " << (long long)(st.bounds_ms + st.combine_ms)
			          << " ms, quantization " << (long long)st.quant_ms << " ms, encode " << (long long)st.encode_ms << " ms, merge " << (long long)st.merge_ms << " ms" << std::endl;
		} else ok(hry_encode(h.cx[0], h.mesh, &o, &out, &out_len));
	} else if (type == "ply") ok(hry_mesh_to_ply(h.mesh, (args.ply_ascii ? HRY_PLY_ASCII : 0) | (args.ply_packed ? HRY_PLY_PACKED : 0), &out, &out_len));
	else ok(hry_mesh_to_obj(h.mesh, 0, &out, &out_len));
	h.bufs.push_back(out);
	write_file(args.out, out, out_len);
	Clock::time_point t3 = Clock::now();
	std::cout << "Writing output took " << ms(t2, t3) << " ms." << std::endl;
	std::cout << "Total compression time: " << ms(t0, t3) << " ms" << std::endl;
	std::cout << "Total input size: " << in.size() << " Bytes" << std::endl;
	std::cout << "Total output size: " << out_len << " Bytes" << std::endl;
	if (trace) std::cerr << "[harry] " << since_start_ms() << " ms  done" << std::endl;
	// Everything is written.  Gigabytes of host arrays, the device contexts and the runtime itself would now be taken apart piece by
	// piece (0.5 s for the configs[3] mesh) only for the process to end: it ends here instead, and the system takes it all back at
	// once.  HRY_ORDERLY_EXIT=1 keeps the long way (leak checkers, tests of the destructors).
	if (!getenv("HRY_ORDERLY_EXIT")) { std::cout.flush(); std::cerr.flush(); fflush(nullptr); _exit(EXIT_SUCCESS); }
	return EXIT_SUCCESS;
}

}   // namespace

int main(int argc, const char **argv)
{
	const Args args = parse(argc, argv);
	try {
		const int rc = run(args);
		if (getenv("HRY_TRACE")) std::cerr << "[harry] " << since_start_ms() << " ms  handles released" << std::endl;
		return rc;
	} catch (const std::exception &e) {
		// what the reference's uncaught exception prints through std::terminate, and the status abort() leaves
		std::cout.flush();
		std::cerr << "terminate called after throwing an instance of 'std::runtime_error'" << std::endl << "  what():  " << e.what() << std::endl;
		return 134;
	}
}
