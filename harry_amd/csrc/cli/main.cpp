// harry -- the reference's command line (main.cc:23-123) on top of the C ABI of libharry_amd.so (include/harry_amd.h):
//     harry [OPTIONS] INPUT OUTPUT      -f/--format hry|ply|obj   -l/--list L   -a/--attr A   -q/--quant Q   -c/--clear-quant
//                                       --ply-ascii   -h/--help
// Same flag state machine (-l sets the list, -a the component for the next -q only, main.cc:47-71), same phase prints
// (main.cc:99-120), same error texts; an error ends the program the way the reference's uncaught std::runtime_error does
// (message of std::terminate on stderr, status 134).  Input kind by magic number, output kind by extension
// (formats/unified_reader.h:33-56, unified_writer.h:31-47): .hry, .ply and .obj both ways.
// Additive options (the reference rejects them as invalid): --profile compat|chunked (default compat = the reference's own
// v0.1 stream), --chunk N, --device D, --shards N (chunked: code the mesh as N shards, one after the other on this GPU, and
// merge them into one sharded container -- what N ranks do in parallel, see harry_amd/sharding.py), --ply-packed (binary PLY
// of a quantised mesh with every value in the width its header declares; the reference's writer dumps the original-width
// records, formats/ply/writer.cc:72-75).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/harry_amd.h"

namespace {

struct QuantArg { int l, o, q; };
struct Args {
	std::string in, out, fmt, profile = "compat";
	std::vector<QuantArg> quant;
	bool clearquant = false, ply_ascii = false, ply_packed = false;
	int chunk = 0, device = 0, shards = 0;
};

struct Opt { char s; const char *l; const char *descr; bool has_val; };
const Opt kOpts[] = {
	{ 'h', "help", "Print this dialogue.", false }, { 'f', "format", "Enforce output format", true }, { 'l', "list", "Select attribute list", true },
	{ 'a', "attr", "Select attribute", true }, { 'q', "quant", "Quantization bits", true }, { 'c', "clear-quant", "Clear all quantization first", false },
	{ 0, "ply-ascii", "PLY writer: Use ASCII format", false }, { 0, "ply-packed", "PLY writer: quantised values in their declared width", false }, { 0, "profile", "compat (reference stream, default) or chunked", true },
	{ 0, "chunk", "chunked: symbols per chunk", true }, { 0, "device", "GPU index", true }, { 0, "shards", "chunked: code as N shards and merge", true } };

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
	}
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

struct Handles {   // released on every path
	hry_ctx *cx = nullptr;
	hry_mesh *mesh = nullptr;
	std::vector<hry_mesh*> shards;
	hry_plan *plan = nullptr;
	std::vector<uint8_t*> bufs;
	~Handles()
	{
		for (uint8_t *b : bufs) hry_free(b);
		for (hry_mesh *s : shards) hry_mesh_free(s);
		if (plan) hry_plan_free(plan);
		if (mesh) hry_mesh_free(mesh);
		if (cx) hry_ctx_destroy(cx);
	}
};

int run(const Args &args)
{
	typedef std::chrono::high_resolution_clock Clock;
	auto ms = [](Clock::time_point a, Clock::time_point b) { return (long long)std::chrono::duration_cast<std::chrono::milliseconds>(b - a).count(); };
	Handles h;
	ok(hry_ctx_create(args.device, &h.cx));

	std::cout << "Reading input..." << std::endl;
	Clock::time_point t0 = Clock::now();
	std::vector<uint8_t> in;
	{
		std::ifstream is(args.in, std::ifstream::binary);
		is.seekg(0, std::ios::end);
		std::streamoff size = is.tellg();
		is.seekg(0, std::ios::beg);
		if (size > 0) { in.resize((size_t)size); is.read((char*)in.data(), size); }
	}
	if (in.size() >= 4 && in[0] == 0xfa && in[1] == 0xff && in[2] == 0xaf && in[3] == 0xaf) ok(hry_decode(h.cx, in.data(), in.size(), nullptr, &h.mesh));
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
	Clock::time_point t1 = Clock::now();
	std::cout << "Reading input took " << ms(t0, t1) << " ms." << std::endl;

	if (!args.quant.empty() || args.clearquant) {
		std::cout << "Quantization..." << std::endl;
		std::vector<hry_quant> q;
		for (const QuantArg &x : args.quant) q.push_back(hry_quant{ x.l, x.o, x.q });
		ok(hry_requant(h.cx, h.mesh, q.data(), q.size(), args.clearquant ? 1 : 0));   // validation and texts of main.cc:74-91 inside
	}
	Clock::time_point t2 = Clock::now();
	if (!args.quant.empty() || args.clearquant) std::cout << "Quantization took " << ms(t1, t2) << " ms." << std::endl;

	std::cout << "Writing output..." << std::endl;
	std::string type = args.fmt;
	if (type.empty()) {
		const std::string e = ext_of(args.out);
		type = e == ".hry" ? "hry" : e == ".ply" ? "ply" : e == ".obj" ? "obj" : "";
		if (type.empty()) throw std::runtime_error("Unknown file extension");
	}
	uint8_t *out = nullptr;
	size_t out_len = 0;
	if (type == "hry") {
		hry_opts o{};
		o.profile = args.profile == "chunked" ? HRY_PROFILE_CHUNKED : HRY_PROFILE_COMPAT;
		o.chunk_syms = args.chunk;
		if (args.shards > 1 && o.profile == HRY_PROFILE_CHUNKED) {
			// the sharded path on one GPU: plan, extract, code every shard, merge.  The whole mesh's bounds come from the mesh itself
			// here (hry_shard_extract copies them); N ranks combine their shards' bounds instead (harry_amd/sharding.py).
			bool need_bounds = false;
			for (int l = 0; l < hry_mesh_nlists(h.mesh); ++l) if (hry_list_ncomp(h.mesh, l) > 0 && !hry_list_min(h.mesh, l)) need_bounds = true;
			if (need_bounds) ok(hry_bounds(h.cx, h.mesh));
			ok(hry_shard_plan(h.mesh, args.shards, &h.plan));
			std::vector<const uint8_t*> parts;
			std::vector<size_t> sizes;
			for (int s = 0; s < args.shards; ++s) {
				hry_mesh *sh = nullptr;
				ok(hry_shard_extract(h.mesh, h.plan, s, &sh));
				h.shards.push_back(sh);
				uint8_t *p = nullptr;
				size_t n = 0;
				ok(hry_encode(h.cx, sh, &o, &p, &n));
				h.bufs.push_back(p);
				parts.push_back(p); sizes.push_back(n);
			}
			ok(hry_merge(parts.data(), sizes.data(), parts.size(), &out, &out_len));
		} else ok(hry_encode(h.cx, h.mesh, &o, &out, &out_len));
	} else if (type == "ply") ok(hry_mesh_to_ply(h.mesh, (args.ply_ascii ? HRY_PLY_ASCII : 0) | (args.ply_packed ? HRY_PLY_PACKED : 0), &out, &out_len));
	else ok(hry_mesh_to_obj(h.mesh, 0, &out, &out_len));
	h.bufs.push_back(out);
	{
		std::ofstream os(args.out, std::ofstream::binary);
		os.write((const char*)out, (std::streamsize)out_len);
		os.flush();
		if (!os) throw std::runtime_error("cannot write " + args.out);
	}
	Clock::time_point t3 = Clock::now();
	std::cout << "Writing output took " << ms(t2, t3) << " ms." << std::endl;
	std::cout << "Total compression time: " << ms(t0, t3) << " ms" << std::endl;
	std::cout << "Total input size: " << in.size() << " Bytes" << std::endl;
	std::cout << "Total output size: " << out_len << " Bytes" << std::endl;
	return EXIT_SUCCESS;
}

}   // namespace

int main(int argc, const char **argv)
{
	const Args args = parse(argc, argv);
	try {
		return run(args);
	} catch (const std::exception &e) {
		// what the reference's uncaught exception prints through std::terminate, and the status abort() leaves
		std::cout.flush();
		std::cerr << "terminate called after throwing an instance of 'std::runtime_error'" << std::endl << "  what():  " << e.what() << std::endl;
		return 134;
	}
}
