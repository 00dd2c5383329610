// mitsuba/dj_merl.cpp -- Mitsuba 0.5 BSDF plugin "dj_merl" on top of the MI355X engine.
//
// Same plugin name, XML properties ("filename") and BSDF method signatures as the reference's
// shell (jdupuy/dj_brdf mitsuba/dj_merl.cpp:18-140): eval = merl evalp, sample/pdf = GGX lobe whose
// roughness is fitted at load time with tabular(merl, 90, /*shadow*/false).  What changes is where
// the arithmetic runs: the MERL table and the fitted lobe live in HBM behind libdjb_hip.so, and
// the fit is one launch of the power-iteration kernel.
//
// NOT COMPILED IN THIS REPOSITORY: it needs the Mitsuba 0.5 SDK (<mitsuba/render/bsdf.h> ...),
// which is absent from the build image.  Copy into mitsuba/src/bsdfs/, add
//   plugins += env.SharedLibrary('dj_merl', ['dj_merl.cpp'], LIBS=['djb_hip'], CPPPATH=[...include])
// to src/bsdfs/SConscript, and see INTEGRATION.md for the wavefront (batched) integration, which
// is the form that actually uses the GPU; the per-intersection calls below are batches of one.
#include <mitsuba/core/fresolver.h>
#include <mitsuba/render/bsdf.h>
#include <mitsuba/hw/basicshader.h>

#include "djb_hip.hpp"

MTS_NAMESPACE_BEGIN

class dj_merl : public BSDF {
public:
	dj_merl(const Properties &props) : BSDF(props), m_brdf(NULL), m_ggx(NULL) {
		ref<FileResolver> fResolver = Thread::getThread()->getFileResolver();
		fs::path path = fResolver->resolve(props.getString("filename"));
		m_brdf = new djb::merl(path.string().c_str());            // throws djb::exc like the reference
		djb::tabular tab(*m_brdf, 90, false);                      // GPU fit
		m_params = djb::tabular::fit_ggx_parameters(tab);
		m_ggx = new djb::ggx();
	}
	dj_merl(Stream *stream, InstanceManager *manager) : BSDF(stream, manager), m_brdf(NULL), m_ggx(NULL) {
		configure();
	}
	~dj_merl() { delete m_brdf; delete m_ggx; }

	void configure() {
		m_components.clear();
		m_components.push_back(EGlossyReflection | EFrontSide | 0);
		m_usesRayDifferentials = false;
		BSDF::configure();
	}

	// djb's i is the light direction and o the viewer: Mitsuba's wi/wo are swapped (dj_brdf.h:23-26)
	Spectrum eval(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (!(bRec.typeMask & EGlossyReflection) || measure != ESolidAngle
			|| Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0)
			return Spectrum(0.0f);
		djb::vec3 o(bRec.wi.x, bRec.wi.y, bRec.wi.z), i(bRec.wo.x, bRec.wo.y, bRec.wo.z);
		djb::vec3 fr_p = m_brdf->evalp(i, o);
		Spectrum s; s.fromLinearRGB(fr_p.x, fr_p.y, fr_p.z);
		return s;
	}

	Float pdf(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (!(bRec.typeMask & EGlossyReflection) || measure != ESolidAngle
			|| Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0)
			return 0.0f;
		djb::vec3 o(bRec.wi.x, bRec.wi.y, bRec.wi.z), i(bRec.wo.x, bRec.wo.y, bRec.wo.z);
		return m_ggx->pdf(i, o, &m_params);
	}

	Spectrum sample(BSDFSamplingRecord &bRec, Float &pdf_, const Point2 &sample) const {
		if (!(bRec.typeMask & EGlossyReflection) || Frame::cosTheta(bRec.wi) <= 0)
			return Spectrum(0.0f);
		djb::vec3 o(bRec.wi.x, bRec.wi.y, bRec.wi.z);
		djb::vec3 i = m_ggx->sample(sample.x, sample.y, o, &m_params);
		if (i.z <= 0) return Spectrum(0.0f);
		bRec.wo = Vector(i.x, i.y, i.z);
		bRec.eta = 1.0f;
		bRec.sampledComponent = 0;
		bRec.sampledType = EGlossyReflection;
		pdf_ = m_ggx->pdf(i, o, &m_params);
		if (pdf_ <= 0) return Spectrum(0.0f);
		return eval(bRec, ESolidAngle) / pdf_;
	}
	Spectrum sample(BSDFSamplingRecord &bRec, const Point2 &sample) const {
		Float pdf_;
		return dj_merl::sample(bRec, pdf_, sample);
	}

	void serialize(Stream *stream, InstanceManager *manager) const { BSDF::serialize(stream, manager); }
	Float getRoughness(const Intersection &its, int component) const {
		float a1, a2; m_params.get_ellipse(&a1, &a2); return 0.5f * (a1 + a2);
	}
	std::string toString() const { return "dj_merl[engine = libdjb_hip (MI355X)]"; }
	Shader *createShader(Renderer *renderer) const;
	MTS_DECLARE_CLASS()
private:
	djb::brdf *m_brdf;
	djb::ggx *m_ggx;
	djb::microfacet::params m_params;
};

// the VPL preview shader is renderer UI, not djb math: a constant diffuse stand-in
class dj_merl_shader : public Shader {
public:
	dj_merl_shader(Renderer *renderer) : Shader(renderer, EBSDFShader) {}
	void generateCode(std::ostringstream &oss, const std::string &evalName,
			const std::vector<std::string> &depNames) const {
		oss << "vec3 " << evalName << "(vec2 uv, vec3 wi, vec3 wo) {\n"
			<< "    if (cosTheta(wi) < 0.0 || cosTheta(wo) < 0.0) return vec3(0.0);\n"
			<< "    return vec3(0.5 * inv_pi * cosTheta(wo));\n}\n\n"
			<< "vec3 " << evalName << "_diffuse(vec2 uv, vec3 wi, vec3 wo) {\n"
			<< "    return " << evalName << "(uv, wi, wo);\n}\n";
	}
	MTS_DECLARE_CLASS()
};
Shader *dj_merl::createShader(Renderer *renderer) const { return new dj_merl_shader(renderer); }

MTS_IMPLEMENT_CLASS(dj_merl_shader, false, Shader)
MTS_IMPLEMENT_CLASS_S(dj_merl, false, BSDF)
MTS_EXPORT_PLUGIN(dj_merl, "dj_merl BRDF (MI355X engine)")
MTS_NAMESPACE_END
