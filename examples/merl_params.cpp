// examples/merl_params.cpp -- Beckmann / GGX roughness of MERL materials, fitted on the GPU.
//
// Counterpart of the reference's driver (jdupuy/dj_brdf examples/merl_params.cpp:30-73) written
// against the djb:: facade of this repository (include/djb_hip.hpp): same command line, same
// params.txt ("# MERL Beckmann GGX", then "name %.3f %.3f" per input, in input order).
// Build: make -C examples      Run: ./merl_params a.binary b.binary ...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
	printf("Usage\n  %s merl1.binary merl2.binary ...\n\n", app);
	printf("Options\n  -h\n     Print help\n\n");
}

} // namespace

int main(int argc, char **argv)
{
	if (argc < 2) { usage(argv[0]); return EXIT_SUCCESS; }
	for (int i = 1; i < argc; ++i)
		if (!strcmp("-h", argv[i])) { usage(argv[0]); return EXIT_SUCCESS; }

	struct row { std::string name; float beckmann, ggx; };
	std::vector<row> rows;
	try {
		for (int i = 1; i < argc; ++i) {
			djb::merl merl(argv[i]);                       // upload + float4 table in HBM
			djb::tabular tab(merl, 90);                    // power-iteration fit kernel
			row r; float dummy;
			r.name = material_name(argv[i]);
			djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&r.beckmann, &dummy, NULL);
			djb::tabular::fit_ggx_parameters(tab).get_ellipse(&r.ggx, &dummy, NULL);
			rows.push_back(r);
		}
	} catch (const djb::exc &e) {
		fprintf(stderr, "%s", e.what());
		return EXIT_FAILURE;
	}

	FILE *pf = fopen("params.txt", "w");
	if (!pf) { perror("params.txt"); return EXIT_FAILURE; }
	fprintf(pf, "# MERL Beckmann GGX\n");
	for (size_t k = 0; k < rows.size(); ++k)
		fprintf(pf, "%s %.3f %.3f\n", rows[k].name.c_str(), rows[k].beckmann, rows[k].ggx);
	fclose(pf);
	return EXIT_SUCCESS;
}
