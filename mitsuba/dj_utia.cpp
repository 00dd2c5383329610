// mitsuba/dj_utia.cpp -- Mitsuba 0.5 BSDF plugin "dj_utia" on top of the MI355X engine (drop-in for jdupuy/dj_brdf
// mitsuba/dj_utia.cpp:16-176; see mitsuba/djb_mitsuba.hpp).
//
// eval = utia::evalp (16-tap interpolation of the UTIA table resident in HBM); sampling is Mitsuba's cosine hemisphere.
// Reference quirks kept: sample() returns M_PI * eval(wi, wo) with Mitsuba's wi / wo passed UNSWAPPED (l.78-82, 95-99)
// whereas eval() swaps them (l.54-56); the estimator is eval * pi, not evalp / pdf; the product M_PI * Color3 is a float
// product (M_PI narrows to Float at Mitsuba's operator*); the constructor prints "Loading <file>" (l.27).
#include "djb_mitsuba.hpp"

MTS_NAMESPACE_BEGIN
using namespace djb_mts;

class dj_utia : public BSDF {
public:
	dj_utia(const Properties &props) : BSDF(props), m_brdf(NULL) {
		m_reflectance = reflectance_property(props);
		const std::string file = resolved(props.getString("filename")).string();
		printf("Loading %s\n", file.c_str());
		m_brdf = new djb::utia(file.c_str());
	}
	dj_utia(Stream *stream, InstanceManager *manager) : BSDF(stream, manager), m_brdf(NULL) { configure(); }
	~dj_utia() { delete m_brdf; }

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
		return warp::squareToCosineHemispherePdf(bRec.wo);
	}
	Spectrum sample(BSDFSamplingRecord &bRec, const Point2 &sample) const {
		if (lobe_masked(bRec, EDiffuseReflection) || at_or_below(bRec.wi))
			return Spectrum(0.0f);
		return cosine_sample(bRec, sample);
	}
	Spectrum sample(BSDFSamplingRecord &bRec, Float &pdf, const Point2 &sample) const {
		if (lobe_masked(bRec, EDiffuseReflection) || at_or_below(bRec.wi))
			return Spectrum(0.0f);                   // l.86-87: pdf is left untouched on this path
		const Spectrum value = cosine_sample(bRec, sample);
		pdf = warp::squareToCosineHemispherePdf(bRec.wo);
		return value;
	}

	void addChild(const std::string &name, ConfigurableObject *child) {
		if (!is_reflectance_child(name, child))
			BSDF::addChild(name, child);
	}
	void serialize(Stream *stream, InstanceManager *manager) const { BSDF::serialize(stream, manager); }
	Float getRoughness(const Intersection &its, int component) const { return std::numeric_limits<Float>::infinity(); }
	std::string toString() const { return id_only("dj_utia", getID()); }
	Shader *createShader(Renderer *renderer) const;
	MTS_DECLARE_CLASS()
private:
	bool unwanted(const BSDFSamplingRecord &bRec, EMeasure measure) const
	{ return lobe_masked(bRec, EDiffuseReflection) || measure != ESolidAngle || at_or_below(bRec.wi) || at_or_below(bRec.wo); }
	// l.76-82 / 89-99: wo from the warp, then pi * eval(wi, wo) -- eval, not evalp, and no swap
	Spectrum cosine_sample(BSDFSamplingRecord &bRec, const Point2 &sample) const {
		bRec.wo = warp::squareToCosineHemisphere(sample);
		bRec.eta = 1.0f;
		bRec.sampledComponent = 0;
		bRec.sampledType = EDiffuseReflection;
		return M_PI * rgb(m_brdf->eval(dir(bRec.wi), dir(bRec.wo)));
	}
	ref<const Texture> m_reflectance;
	djb::utia *m_brdf;
};

DJB_MTS_PREVIEW_SHADER(dj_utia_shader)
Shader *dj_utia::createShader(Renderer *renderer) const { return new dj_utia_shader(renderer, m_reflectance.get()); }

MTS_IMPLEMENT_CLASS(dj_utia_shader, false, Shader)
MTS_IMPLEMENT_CLASS_S(dj_utia, false, BSDF)
MTS_EXPORT_PLUGIN(dj_utia, "dj_utia BRDF")
MTS_NAMESPACE_END
