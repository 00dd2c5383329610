// mitsuba/dj_abc.cpp -- Mitsuba 0.5 BSDF plugin "dj_abc" (and, with MODEL = sgd, "dj_sgd": see
// mitsuba/dj_sgd.cpp) on top of the MI355X engine.
//
// Same plugin name, "material" property and BSDF signatures as the reference's shells
// (jdupuy/dj_brdf mitsuba/dj_abc.cpp:19-141, mitsuba/dj_sgd.cpp:19-141): eval = model evalp;
// sample/pdf = a tabulated lobe fitted at load time, djb::tabular(model, 90), which samples with
// the non-VNDF "nmap" scheme (supports_smith_vndf_sampling() == false).
// NOT COMPILED HERE (no Mitsuba SDK in the image); see mitsuba/dj_merl.cpp and INTEGRATION.md.
#include <mitsuba/core/fresolver.h>
#include <mitsuba/render/bsdf.h>

#include "djb_hip.hpp"

#ifndef DJ_MODEL
#	define DJ_MODEL abc
#	define DJ_PLUGIN dj_abc
#	define DJ_PLUGIN_STR "dj_abc"
#endif

MTS_NAMESPACE_BEGIN

class DJ_PLUGIN : public BSDF {
public:
	DJ_PLUGIN(const Properties &props) : BSDF(props), m_brdf(NULL), m_tab(NULL) {
		// the reference runs the material NAME through the file resolver (dj_abc.cpp:29-30); kept
		ref<FileResolver> fResolver = Thread::getThread()->getFileResolver();
		fs::path name = fResolver->resolve(props.getString("material"));
		m_brdf = new djb::DJ_MODEL(name.filename().string().c_str());   // djb::exc on unknown material
		m_tab = new djb::tabular(*m_brdf, 90);                          // GPU power-iteration fit
	}
	DJ_PLUGIN(Stream *stream, InstanceManager *manager) : BSDF(stream, manager), m_brdf(NULL), m_tab(NULL) {
		configure();
	}
	~DJ_PLUGIN() { delete m_tab; delete m_brdf; }

	void configure() {
		m_components.clear();
		m_components.push_back(EGlossyReflection | EFrontSide | 0);
		m_usesRayDifferentials = false;
		BSDF::configure();
	}

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
		return m_tab->pdf(i, o);
	}
	Spectrum sample(BSDFSamplingRecord &bRec, Float &pdf_, const Point2 &sample) const {
		if (!(bRec.typeMask & EGlossyReflection) || Frame::cosTheta(bRec.wi) <= 0)
			return Spectrum(0.0f);
		djb::vec3 o(bRec.wi.x, bRec.wi.y, bRec.wi.z);
		djb::vec3 i = m_tab->sample(sample.x, sample.y, o);
		if (i.z <= 0) return Spectrum(0.0f);
		bRec.wo = Vector(i.x, i.y, i.z);
		bRec.eta = 1.0f;
		bRec.sampledComponent = 0;
		bRec.sampledType = EGlossyReflection;
		pdf_ = m_tab->pdf(i, o);
		if (pdf_ <= 0) return Spectrum(0.0f);
		return eval(bRec, ESolidAngle) / pdf_;
	}
	Spectrum sample(BSDFSamplingRecord &bRec, const Point2 &sample) const {
		Float pdf_;
		return DJ_PLUGIN::sample(bRec, pdf_, sample);
	}
	void serialize(Stream *stream, InstanceManager *manager) const { BSDF::serialize(stream, manager); }
	Float getRoughness(const Intersection &its, int component) const { return 0.5f; }
	std::string toString() const { return DJ_PLUGIN_STR "[engine = libdjb_hip (MI355X)]"; }
	MTS_DECLARE_CLASS()
private:
	djb::brdf *m_brdf;
	djb::tabular *m_tab;
};

MTS_IMPLEMENT_CLASS_S(DJ_PLUGIN, false, BSDF)
MTS_EXPORT_PLUGIN(DJ_PLUGIN, DJ_PLUGIN_STR " BRDF (MI355X engine)")
MTS_NAMESPACE_END
