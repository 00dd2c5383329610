// tests/mitsuba_mock/shell_harness.cpp -- TEST INFRASTRUCTURE.  One Mitsuba BSDF shell + the functional mock of
// mitsuba/mock.h compiled into one shared object with a C interface that tests/mitsuba_mock/shell_cases.py drives via ctypes.
//
//   g++ -std=c++11 -O2 -fPIC -shared -DMITSUBA_MOCK_MAIN -DSHELL_SOURCE='"<shell>.cpp"' -I tests/mitsuba_mock <side includes> ...
//
// reference side (build container only): SHELL_SOURCE = /root/reference/mitsuba/<shell>.cpp, -I /root/reference
//                                        -> oracle/_ref/shells/libshell_<shell>.so  (git-ignored)
// repository side:                       SHELL_SOURCE = mitsuba/<shell>.cpp, -I include -I mitsuba, -ldjb_hip
//                                        -> tests/mitsuba_mock/_build/libshell_<shell>.so
// The shell source is included textually: the plugin's exported CreateInstance is the only way in, as in Mitsuba.
#include SHELL_SOURCE

#include <exception>

using namespace mitsuba;

namespace {
void put(char *dst, int cap, const std::string &s)
{
	if (!dst || cap <= 0) return;
	size_t n = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
	memcpy(dst, s.data(), n); dst[n] = 0;
}
std::string joined(const std::vector<std::string> &v)
{
	std::string r;
	for (size_t k = 0; k < v.size(); ++k) { if (k) r += "\n"; r += v[k]; }
	return r;
}
#define HARNESS_TRY try {
#define HARNESS_CATCH(err, cap) } catch (const std::exception &e) { put(err, cap, std::string("exception: ") + e.what()); return 1; } \
	catch (...) { put(err, cap, "exception: unknown"); return 1; } put(err, cap, ""); return 0;
}

extern "C" {

// ---- scene description
void *props_new(const char *id) { Properties *p = new Properties(); p->id = id; return p; }
void props_set_string(void *p, const char *n, const char *v) { Properties::Value x; x.type = Properties::EString; x.str = v; x.f = 0; x.b = false; ((Properties *)p)->values[n] = x; }
void props_set_float(void *p, const char *n, float v) { Properties::Value x; x.type = Properties::EFloat; x.f = v; x.b = false; ((Properties *)p)->values[n] = x; }
void props_set_boolean(void *p, const char *n, int v) { Properties::Value x; x.type = Properties::EBoolean; x.f = 0; x.b = v != 0; ((Properties *)p)->values[n] = x; }
void props_set_spectrum(void *p, const char *n, float r, float g, float b) { Properties::Value x; x.type = Properties::ESpectrum; x.f = 0; x.b = false; x.spec = Spectrum(r, g, b); ((Properties *)p)->values[n] = x; }
void props_queried(void *p, char *out, int cap) { put(out, cap, joined(((Properties *)p)->queried)); }
void resolver_log(char *out, int cap) { put(out, cap, joined(FileResolver::resolved())); FileResolver::resolved().clear(); }

// ---- plugin life cycle: constructor (+ children) + configure, as the scene loader does
int shell_create(void *props, void **out, char *err, int cap)
{
	HARNESS_TRY
	*out = CreateInstance((const Properties *)props);
	HARNESS_CATCH(err, cap)
}
// kind 0: a texture  value(u, v) = a + bu * u + cv * v  (coef = a.rgb, bu.rgb, cv.rgb); kind 1: a non-texture child
int shell_add_child(void *bsdf, const char *name, int kind, const float *c, char *err, int cap)
{
	HARNESS_TRY
	ConfigurableObject *child = kind == 0
		? (ConfigurableObject *)new AffineTexture(Spectrum(c[0], c[1], c[2]), Spectrum(c[3], c[4], c[5]), Spectrum(c[6], c[7], c[8]))
		: new ConfigurableObject();
	((BSDF *)bsdf)->addChild(name, child);
	HARNESS_CATCH(err, cap)
}
int shell_configure(void *bsdf, char *err, int cap)
{
	HARNESS_TRY
	((BSDF *)bsdf)->configure();
	HARNESS_CATCH(err, cap)
}
// components (up to cap_c), their count, usesRayDifferentials, the energy-conservation requests, toString
int shell_info(void *bsdf_, unsigned int *components, int cap_c, int *n_components, int *uses_rd, char *ensured, int cap_e,
               char *str, int cap_s, char *err, int cap)
{
	HARNESS_TRY
	BSDF *bsdf = (BSDF *)bsdf_;
	*n_components = (int)bsdf->m_components.size();
	for (int k = 0; k < *n_components && k < cap_c; ++k) components[k] = bsdf->m_components[(size_t)k];
	*uses_rd = bsdf->m_usesRayDifferentials ? 1 : 0;
	put(ensured, cap_e, joined(bsdf->m_ensured));
	put(str, cap_s, bsdf->toString());
	HARNESS_CATCH(err, cap)
}
float shell_roughness(void *bsdf, float u, float v, int component)
{
	Intersection its; its.u = u; its.v = v;
	return ((BSDF *)bsdf)->getRoughness(its, component);
}

// ---- the three BSDF queries over n records.  rec layout per record: wi[3], wo[3], uv[2]; masks[n], components[n], measures[n]
int shell_eval(void *bsdf, int n, const float *wi, const float *wo, const float *uv, const unsigned int *mask,
               const int *component, const int *measure, float *out_rgb, float *out_pdf, char *err, int cap)
{
	HARNESS_TRY
	for (int k = 0; k < n; ++k) {
		Intersection its; its.u = uv[2 * k]; its.v = uv[2 * k + 1];
		BSDFSamplingRecord bRec(its);
		bRec.wi = Vector(wi[3 * k], wi[3 * k + 1], wi[3 * k + 2]);
		bRec.wo = Vector(wo[3 * k], wo[3 * k + 1], wo[3 * k + 2]);
		bRec.typeMask = mask[k]; bRec.component = component[k];
		Spectrum s = ((BSDF *)bsdf)->eval(bRec, (EMeasure)measure[k]);
		s.toLinearRGB(out_rgb[3 * k], out_rgb[3 * k + 1], out_rgb[3 * k + 2]);
		out_pdf[k] = ((BSDF *)bsdf)->pdf(bRec, (EMeasure)measure[k]);
	}
	HARNESS_CATCH(err, cap)
}
// with_pdf: the 3-argument overload.  out_meta per record: eta, sampledComponent, sampledType.  Fields the shell leaves
// alone keep their sentinels (wo = 0, eta = -7, sampledComponent = -7, sampledType = 0xDEAD, pdf = -7).
int shell_sample(void *bsdf, int n, const float *wi, const float *uv, const float *xi, const unsigned int *mask,
                 const int *component, int with_pdf, float *out_rgb, float *out_pdf, float *out_wo, float *out_meta,
                 char *err, int cap)
{
	HARNESS_TRY
	for (int k = 0; k < n; ++k) {
		Intersection its; its.u = uv[2 * k]; its.v = uv[2 * k + 1];
		BSDFSamplingRecord bRec(its);
		bRec.wi = Vector(wi[3 * k], wi[3 * k + 1], wi[3 * k + 2]);
		bRec.typeMask = mask[k]; bRec.component = component[k];
		bRec.eta = -7.0f; bRec.sampledComponent = -7; bRec.sampledType = 0xDEADu;
		Float pdf = -7.0f;
		Point2 sample(xi[2 * k], xi[2 * k + 1]);
		Spectrum s = with_pdf ? ((BSDF *)bsdf)->sample(bRec, pdf, sample) : ((BSDF *)bsdf)->sample(bRec, sample);
		s.toLinearRGB(out_rgb[3 * k], out_rgb[3 * k + 1], out_rgb[3 * k + 2]);
		out_pdf[k] = pdf;
		out_wo[3 * k] = bRec.wo.x; out_wo[3 * k + 1] = bRec.wo.y; out_wo[3 * k + 2] = bRec.wo.z;
		out_meta[3 * k] = bRec.eta; out_meta[3 * k + 1] = (float)bRec.sampledComponent; out_meta[3 * k + 2] = (float)bRec.sampledType;
	}
	HARNESS_CATCH(err, cap)
}

// ---- serialization: serialize -> stream size and instance count; optionally rebuild through the unserializing constructor
// and report the clone's toString / components (the clone is never destroyed: the reference's unserializing constructors leave
// their djb pointers uninitialised)
int shell_serialize(void *bsdf_, int rebuild, int *n_bytes, int *n_instances, unsigned int *clone_components, char *clone_str,
                    int cap_s, char *err, int cap)
{
	HARNESS_TRY
	BSDF *bsdf = (BSDF *)bsdf_;
	Stream stream; InstanceManager manager;
	bsdf->serialize(&stream, &manager);
	*n_bytes = (int)stream.bytes.size(); *n_instances = (int)manager.stored.size();
	if (rebuild) {
		BSDF *clone = (BSDF *)CreateInstanceFromStream(&stream, &manager);
		*clone_components = clone->m_components.empty() ? 0u : clone->m_components[0];
		put(clone_str, cap_s, clone->toString());
		if (stream.pos != stream.bytes.size()) throw std::runtime_error("unserialize left bytes in the stream");
	}
	HARNESS_CATCH(err, cap)
}

// ---- VPL preview shader: GLSL text, dependency count, completeness, uniform names / values, renderer bookkeeping
int shell_shader(void *bsdf, char *code, int cap_c, int *n_deps, int *complete, char *uniforms, int cap_u, float *values9,
                 int *registered, int *unregistered, char *err, int cap)
{
	HARNESS_TRY
	Renderer renderer;
	Shader *shader = ((BSDF *)bsdf)->createShader(&renderer);
	if (!shader) { put(code, cap_c, ""); *n_deps = -1; *complete = 0; put(uniforms, cap_u, ""); *registered = *unregistered = 0; }
	else {
		std::vector<Shader *> deps; shader->putDependencies(deps);
		*n_deps = (int)deps.size(); *complete = shader->isComplete() ? 1 : 0;
		std::vector<std::string> names;
		for (size_t k = 0; k < deps.size(); ++k) { char b[16]; snprintf(b, sizeof b, "dep%d", (int)k); names.push_back(b); }
		std::ostringstream oss; shader->generateCode(oss, "bsdf0", names);
		put(code, cap_c, oss.str());
		GPUProgram program; std::vector<int> ids; int unit = 0;
		shader->resolve(&program, "bsdf0", ids); shader->bind(&program, ids, unit);
		put(uniforms, cap_u, joined(program.names));
		for (size_t k = 0; k < program.values.size() && k < 3; ++k) for (int c = 0; c < 3; ++c) values9[3 * k + c] = program.values[k].s[c];
		shader->cleanup(&renderer);
		*registered = renderer.registered; *unregistered = renderer.unregistered;
	}
	HARNESS_CATCH(err, cap)
}

} // extern "C"
