// mitsuba/dj_merl.cpp -- Mitsuba 0.5 BSDF plugin "dj_merl" on top of the MI355X engine (drop-in for jdupuy/dj_brdf
// mitsuba/dj_merl.cpp:17-189; see mitsuba/djb_mitsuba.hpp for what "drop-in" covers and how it is tested).
//
// eval = merl::evalp; sample / pdf = a GGX lobe whose roughness is fitted at load time with tabular(merl, 90, shadow=false)
// (l.32: the plugin, unlike examples/merl_params.cpp, fits without shadowing).  The table and the fit live in HBM; the fit
// is one launch of the power-iteration kernel.
// Reference quirks kept: the single component is registered as EDiffuseReflection (l.51) and that is what eval / pdf /
// sample test (l.57, 69, 81) although sample reports EGlossyReflection (l.92); no getRoughness override.
#include "djb_mitsuba.hpp"

MTS_NAMESPACE_BEGIN
using namespace djb_mts;

class dj_merl : public BSDF {
public:
	dj_merl(const Properties &props) : BSDF(props), m_brdf(NULL), m_ggx(NULL) {
		m_reflectance = reflectance_property(props);
		const std::string file = resolved(props.getString("filename")).string();
		m_brdf = new djb::merl(file.c_str());                                   // throws djb::exc like the reference
		m_params = djb::tabular::fit_ggx_parameters(djb::tabular(*m_brdf, 90, false));
		m_ggx = new djb::ggx();
	}
	dj_merl(Stream *stream, InstanceManager *manager) : BSDF(stream, manager), m_brdf(NULL), m_ggx(NULL) { configure(); }
	~dj_merl() { delete m_brdf; delete m_ggx; }

	void configure() {
		m_components.clear();
		m_components.push_back(EDiffuseReflection | EFrontSide | 0);
		m_usesRayDifferentials = false;
		BSDF::configure();
	}

	Spectrum eval(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (unwanted(bRec, measure))
			return Spectrum(0.0f);
		return rgb(m_brdf->evalp(dir(bRec.wo), dir(bRec.wi)));
	}
	Float pdf(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (unwanted(bRec, measure))
			return 0.0f;
		return m_ggx->pdf(dir(bRec.wo), dir(bRec.wi), &m_params);
	}
	Spectrum sample(BSDFSamplingRecord &bRec, const Point2 &sample) const {
		if (lobe_masked(bRec, EDiffuseReflection) || at_or_below(bRec.wi))
			return Spectrum(0.0f);
		const djb::vec3 o = dir(bRec.wi);
		return finish_lobe_sample(*this, *m_brdf, bRec, m_ggx->sample(sample.x, sample.y, o, &m_params), o);
	}
	Spectrum sample(BSDFSamplingRecord &bRec, Float &pdf_, const Point2 &sample_) const {
		Spectrum res = sample(bRec, sample_);
		pdf_ = pdf(bRec, ESolidAngle);        // l.101-102: evaluated even when the sample was rejected
		return res;
	}

	void addChild(const std::string &name, ConfigurableObject *child) {
		if (!is_reflectance_child(name, child))      // l.106-113: the texture child is accepted and dropped
			BSDF::addChild(name, child);
	}
	void serialize(Stream *stream, InstanceManager *manager) const { BSDF::serialize(stream, manager); }
	std::string toString() const { return id_only("dj_merl", getID()); }
	Shader *createShader(Renderer *renderer) const;
	MTS_DECLARE_CLASS()
private:
	bool unwanted(const BSDFSamplingRecord &bRec, EMeasure measure) const
	{ return lobe_masked(bRec, EDiffuseReflection) || measure != ESolidAngle || at_or_below(bRec.wi) || at_or_below(bRec.wo); }
	ref<const Texture> m_reflectance;
	djb::brdf *m_brdf;
	djb::ggx *m_ggx;
	djb::microfacet::params m_params;
};

DJB_MTS_PREVIEW_SHADER(dj_merl_shader)
Shader *dj_merl::createShader(Renderer *renderer) const { return new dj_merl_shader(renderer, m_reflectance.get()); }

MTS_IMPLEMENT_CLASS(dj_merl_shader, false, Shader)
MTS_IMPLEMENT_CLASS_S(dj_merl, false, BSDF)
MTS_EXPORT_PLUGIN(dj_merl, "dj_merl BRDF")
MTS_NAMESPACE_END
