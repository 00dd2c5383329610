// examples/merl_params.cpp -- Beckmann / GGX roughness of MERL materials, fitted on the GPU.
//
// Counterpart of the reference's driver (jdupuy/dj_brdf examples/merl_params.cpp:30-73) written
// against the djb:: facade of this repository (include/djb_hip.hpp): same command line, same
// params.txt ("# MERL Beckmann GGX", then "name %.3f %.3f" per input, in input order).
//
// Default mode: the files are dealt round-robin to every visible GPU (one host thread + one
// context/stream per GPU, no exchange between GPUs); each GPU runs the native pipeline
// djb_fit_merl_files (reader threads -> pinned ring -> async upload -> conversion -> ONE fit launch).
// `-s` runs the reference's own loop shape instead (load, fit, next file) on the djb:: classes.
// DJB_EXAMPLE_SHARE_GPU=1 (self-test on a box with fewer GPUs than `-g N` asks for): the N host threads still run, each with
// its own context and stream, but context g lives on device g mod (number of GPUs) -- the multi-GPU code path on one device.
// Build: make -C examples      Run: ./merl_params [-s] [-g N] a.binary b.binary ...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "djb_hip.hpp"

namespace {

// basename up to the first '.', what sscanf(strrchr(input,'/')+1, "%[^.]s", name) extracts
// (reference examples/merl_params.cpp:64); also accepts paths without a '/'.
std::string material_name(const char *path)
{
	const char *slash = strrchr(path, '/');
	std::string base(slash ? slash + 1 : path);
	std::string name = base.substr(0, base.find('.'));
	if (name.size() > 63) name.resize(63);
	return name;
}

void usage(const char *app)
{
	printf("%s - GGX and Beckmann Parameters for MERL BRDFs (MI355X)\n\n", app);
	printf("Usage\n  %s [-s] [-g N] merl1.binary merl2.binary ...\n\n", app);
	printf("Options\n  -h\n     Print help\n  -s\n     One file at a time on the djb:: classes (the reference's loop)\n"
	       "  -g N\n     Use N GPUs (default: all)\n\n");
}

} // namespace

int main(int argc, char **argv)
{
	if (argc < 2) { usage(argv[0]); return EXIT_SUCCESS; }
	bool sequential = false;
	int gpus = 0;
	std::vector<const char *> files;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp("-h", argv[i])) { usage(argv[0]); return EXIT_SUCCESS; }
		else if (!strcmp("-s", argv[i])) sequential = true;
		else if (!strcmp("-g", argv[i]) && i + 1 < argc) gpus = atoi(argv[++i]);
		else files.push_back(argv[i]);
	}
	const int n = (int)files.size();
	std::vector<float> beckmann(n), ggx(n);

	if (sequential) {
		try {
			for (int k = 0; k < n; ++k) {
				djb::merl merl(files[k]);                      // upload + packed RGB table in HBM
				djb::tabular tab(merl, 90);                    // power-iteration fit kernel
				float dummy;
				djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&beckmann[k], &dummy, NULL);
				djb::tabular::fit_ggx_parameters(tab).get_ellipse(&ggx[k], &dummy, NULL);
			}
		} catch (const djb::exc &e) {
			fprintf(stderr, "%s", e.what());
			return EXIT_FAILURE;
		}
	} else {
		int n_dev = djb::hip::context::device_count();
		const bool on_cpu = djb::hip::context::standard_device() == DJB_DEVICE_CPU;   // DJB_DEVICE=cpu, or no HIP device at all
		if (on_cpu) n_dev = 1;                       // one host context; djb_fit_merl_files spreads the files over its threads
		const char *share_env = getenv("DJB_EXAMPLE_SHARE_GPU");
		const bool share = !on_cpu && share_env && !strcmp(share_env, "1") && n_dev > 0;
		if (gpus <= 0 || (gpus > n_dev && !share)) gpus = n_dev;
		if (gpus > n && n > 0) gpus = n;
		// ONE library call for the whole job: file k -> context k mod G, a host thread per context inside the library,
		// rows in input order, no exchange between GPUs
		std::vector<djb_ctx *> ctxs;
		djb_status st = DJB_OK;
		for (int g = 0; g < gpus && st == DJB_OK; ++g) {
			djb_ctx *c = NULL;
			st = djb_ctx_create(on_cpu ? DJB_DEVICE_CPU : share ? g % n_dev : g, &c);
			if (st == DJB_OK) ctxs.push_back(c);
		}
		if (st == DJB_OK && n > 0)
			st = djb_fit_merl_files_multi(&ctxs[0], (int)ctxs.size(), n, &files[0], 90, 1, 0, &beckmann[0], &ggx[0], NULL);
		const std::string error = st != DJB_OK ? djb_last_error() : "";
		for (size_t g = 0; g < ctxs.size(); ++g) djb_ctx_destroy(ctxs[g]);
		if (st != DJB_OK) { fprintf(stderr, "%s", error.c_str()); return EXIT_FAILURE; }
	}

	FILE *pf = fopen("params.txt", "w");
	if (!pf) { perror("params.txt"); return EXIT_FAILURE; }
	fprintf(pf, "# MERL Beckmann GGX\n");
	for (int k = 0; k < n; ++k)
		fprintf(pf, "%s %.3f %.3f\n", material_name(files[k]).c_str(), beckmann[k], ggx[k]);
	fclose(pf);
	return EXIT_SUCCESS;
}
