// mitsuba/dj_utia.cpp -- Mitsuba 0.5 BSDF plugin "dj_utia" on top of the MI355X engine.
//
// Same plugin name, "filename" property and BSDF signatures as the reference's shell
// (jdupuy/dj_brdf mitsuba/dj_utia.cpp:16-133): eval = utia evalp, cosine-hemisphere sampling from
// Mitsuba's warp::.  The reference passes wi/wo UNSWAPPED to eval inside sample() (l.78-80, 95-97)
// while eval() swaps them; UTIA data is not reciprocal-symmetrised, so that quirk is kept.
// NOT COMPILED HERE (no Mitsuba SDK in the image); see mitsuba/dj_merl.cpp and INTEGRATION.md.
#include <mitsuba/core/fresolver.h>
#include <mitsuba/core/warp.h>
#include <mitsuba/render/bsdf.h>

#include "djb_hip.hpp"

MTS_NAMESPACE_BEGIN

class dj_utia : public BSDF {
public:
	dj_utia(const Properties &props) : BSDF(props), m_brdf(NULL) {
		ref<FileResolver> fResolver = Thread::getThread()->getFileResolver();
		m_brdf = new djb::utia(fResolver->resolve(props.getString("filename")).string().c_str());
	}
	dj_utia(Stream *stream, InstanceManager *manager) : BSDF(stream, manager), m_brdf(NULL) { configure(); }
	~dj_utia() { delete m_brdf; }

	void configure() {
		m_components.clear();
		m_components.push_back(EDiffuseReflection | EFrontSide | 0);
		m_usesRayDifferentials = false;
		BSDF::configure();
	}

	Spectrum evalDirs(const Vector &a, const Vector &b) const {
		djb::vec3 fr_p = m_brdf->evalp(djb::vec3(a.x, a.y, a.z), djb::vec3(b.x, b.y, b.z));
		Spectrum s; s.fromLinearRGB(fr_p.x, fr_p.y, fr_p.z);
		return s;
	}
	Spectrum eval(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (!(bRec.typeMask & EDiffuseReflection) || measure != ESolidAngle
			|| Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0)
			return Spectrum(0.0f);
		return evalDirs(/* i = */bRec.wo, /* o = */bRec.wi);
	}
	Float pdf(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (!(bRec.typeMask & EDiffuseReflection) || measure != ESolidAngle
			|| Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0)
			return 0.0f;
		return warp::squareToCosineHemispherePdf(bRec.wo);
	}
	Spectrum sample(BSDFSamplingRecord &bRec, Float &pdf_, const Point2 &sample) const {
		if (!(bRec.typeMask & EDiffuseReflection) || Frame::cosTheta(bRec.wi) <= 0)
			return Spectrum(0.0f);
		bRec.wo = warp::squareToCosineHemisphere(sample);
		bRec.eta = 1.0f;
		bRec.sampledComponent = 0;
		bRec.sampledType = EDiffuseReflection;
		pdf_ = warp::squareToCosineHemispherePdf(bRec.wo);
		return evalDirs(bRec.wi, bRec.wo) / pdf_;          // unswapped, as in the reference
	}
	Spectrum sample(BSDFSamplingRecord &bRec, const Point2 &sample) const {
		Float pdf_;
		return dj_utia::sample(bRec, pdf_, sample);
	}
	void serialize(Stream *stream, InstanceManager *manager) const { BSDF::serialize(stream, manager); }
	Float getRoughness(const Intersection &its, int component) const { return std::numeric_limits<Float>::infinity(); }
	std::string toString() const { return "dj_utia[engine = libdjb_hip (MI355X)]"; }
	MTS_DECLARE_CLASS()
private:
	djb::brdf *m_brdf;
};

MTS_IMPLEMENT_CLASS_S(dj_utia, false, BSDF)
MTS_EXPORT_PLUGIN(dj_utia, "dj_utia BRDF (MI355X engine)")
MTS_NAMESPACE_END
