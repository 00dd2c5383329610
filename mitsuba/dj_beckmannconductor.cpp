// mitsuba/dj_beckmannconductor.cpp -- Mitsuba 0.5 BSDF plugin "dj_beckmannconductor" on top of
// the MI355X engine.
//
// Same plugin name, XML properties (alpha / alpha1 / alpha2 / alphaAngle, material | eta + k,
// leanmap1 / leanmap2, leanFiltering, dmapscale, merl) and BSDF signatures as the reference's
// shell (jdupuy/dj_brdf mitsuba/dj_beckmannconductor.cpp:151-491).  Per hit the reference builds
//     params  = elliptic(alpha1, alpha2, phi)            -> lrep1 (params_to_lrep), lrep1 *= scale
//     lrep2   = LEAN-map texel moments (E1, E2 biased by 25, E5 by 625: l.300-303)
//     params' = lrep_to_params(lrep1 + lrep2)
//     value   = evalp(i, o, &params') * fresnelConductorExact(...)
// Everything up to `value` is one call here (beckmann::evalp_lean / djb_eval_lean_batch); the exact
// conductor Fresnel stays Mitsuba's (its arithmetic lives in the renderer).
// NOT COMPILED HERE (no Mitsuba SDK in the image); see mitsuba/dj_merl.cpp and INTEGRATION.md.
#include <mitsuba/core/fresolver.h>
#include <mitsuba/render/bsdf.h>
#include <mitsuba/render/texture.h>
#include <mitsuba/hw/basicshader.h>
#include "microfacet.h"
#include "ior.h"

#include "djb_hip.hpp"

MTS_NAMESPACE_BEGIN

class dj_beckmann_conductor : public BSDF {
public:
	static const int BIAS = 25;   // the LEAN maps store E1, E2 with a +25 bias (and E5 + 625)

	dj_beckmann_conductor(const Properties &props) : BSDF(props), m_brdf(NULL) {
		ref<FileResolver> fResolver = Thread::getThread()->getFileResolver();
		m_specularReflectance = new ConstantSpectrumTexture(props.getSpectrum("specularReflectance", Spectrum(1.0f)));
		std::string materialName = props.getString("material", "Cu");
		Spectrum intEta, intK;
		if (boost::to_lower_copy(materialName) == "none") {
			intEta = Spectrum(0.0f); intK = Spectrum(1.0f);
		} else {
			intEta.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + materialName + ".eta.spd")));
			intK.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + materialName + ".k.spd")));
		}
		Float extEta = lookupIOR(props, "extEta", "air");
		m_eta = props.getSpectrum("eta", intEta) / extEta;
		m_k = props.getSpectrum("k", intK) / extEta;

		Float alpha = props.getFloat("alpha", 0.1f);
		m_alpha1 = props.getFloat("alpha1", alpha);
		m_alpha2 = props.getFloat("alpha2", alpha);
		m_alphaAngle = degToRad(props.getFloat("alphaAngle", 0.0f));
		m_scale = props.getFloat("dmapscale", 1.0f);
		m_brdf = new djb::beckmann(djb::fresnel::ideal(), true);
		if (props.hasProperty("merl")) {   // base roughness fitted from a MERL file (l.179-190), on the GPU
			djb::merl merl(fResolver->resolve(props.getString("merl")).string().c_str());
			djb::tabular tab(merl, 90, true);
			float a, dummy;
			djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&a, &dummy);
			m_alpha1 = m_alpha2 = a;
		}
		m_leanmap1 = new ConstantFloatTexture(0.0f);
		m_leanmap2 = new ConstantFloatTexture(0.0f);
	}
	dj_beckmann_conductor(Stream *stream, InstanceManager *manager) : BSDF(stream, manager), m_brdf(NULL) { configure(); }
	~dj_beckmann_conductor() { delete m_brdf; }

	void addChild(const std::string &name, ConfigurableObject *child) {
		if (child->getClass()->derivesFrom(MTS_CLASS(Texture))) {
			if (name == "leanmap1") m_leanmap1 = static_cast<Texture *>(child);
			else if (name == "leanmap2") m_leanmap2 = static_cast<Texture *>(child);
			else if (name == "specularReflectance") m_specularReflectance = static_cast<Texture *>(child);
			else BSDF::addChild(name, child);
		} else BSDF::addChild(name, child);
	}

	void configure() {
		m_components.clear();
		m_components.push_back(EGlossyReflection | EFrontSide | ESpatiallyVarying | EAnisotropic);
		m_usesRayDifferentials = true;
		BSDF::configure();
	}

	// texel -> slope moments of the LEAN map (E1, E2 from map 1; E3, E4, E5 from map 2)
	void leanMoments(const Intersection &its, float lean[5]) const {
		Spectrum m1 = m_leanmap1->eval(its, true), m2 = m_leanmap2->eval(its, true);
		Float r1, g1, b1, r2, g2, b2;
		m1.toLinearRGB(r1, g1, b1); m2.toLinearRGB(r2, g2, b2);
		lean[0] = r1 - BIAS; lean[1] = g1 - BIAS;
		lean[2] = r2; lean[3] = g2; lean[4] = b2 - BIAS * BIAS;
	}

	Spectrum eval(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (!(bRec.typeMask & EGlossyReflection) || measure != ESolidAngle
			|| Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0)
			return Spectrum(0.0f);
		djb::vec3 o(bRec.wi.x, bRec.wi.y, bRec.wi.z), i(bRec.wo.x, bRec.wo.y, bRec.wo.z), fr_cos;
		float lean[5];
		leanMoments(bRec.its, lean);
		m_brdf->evalp_lean(1, &i, &o, djb::microfacet::params::elliptic(m_alpha1, m_alpha2, m_alphaAngle),
		                   m_scale, lean, &fr_cos);
		Vector H = normalize(bRec.wo + bRec.wi);
		const Spectrum F = fresnelConductorExact(dot(bRec.wi, H), m_eta, m_k) * m_specularReflectance->eval(bRec.its);
		return F * fr_cos.x;
	}

	Float pdf(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (!(bRec.typeMask & EGlossyReflection) || measure != ESolidAngle
			|| Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0)
			return 0.0f;
		djb::vec3 o(bRec.wi.x, bRec.wi.y, bRec.wi.z), i(bRec.wo.x, bRec.wo.y, bRec.wo.z), fr_cos;
		float lean[5], pdf_;
		leanMoments(bRec.its, lean);
		m_brdf->evalp_lean(1, &i, &o, djb::microfacet::params::elliptic(m_alpha1, m_alpha2, m_alphaAngle),
		                   m_scale, lean, &fr_cos, &pdf_);
		return pdf_;
	}

	Spectrum sample(BSDFSamplingRecord &bRec, Float &pdf_, const Point2 &sample) const {
		if (!(bRec.typeMask & EGlossyReflection) || Frame::cosTheta(bRec.wi) <= 0)
			return Spectrum(0.0f);
		float lean[5];
		leanMoments(bRec.its, lean);
		djb::beckmann::lrep l1, l2(lean[0], lean[1], lean[2], lean[3], lean[4]);
		djb::beckmann::params_to_lrep(djb::microfacet::params::elliptic(m_alpha1, m_alpha2, m_alphaAngle), &l1);
		l1 *= m_scale;
		djb::microfacet::params params;
		djb::beckmann::lrep_to_params(l1 + l2, &params);
		djb::vec3 o(bRec.wi.x, bRec.wi.y, bRec.wi.z), i;
		djb::vec3 w = m_brdf->evalp_is(sample.x, sample.y, o, &i, &pdf_, &params);
		if (pdf_ <= 0 || i.z <= 0) return Spectrum(0.0f);
		bRec.wo = Vector(i.x, i.y, i.z);
		bRec.eta = 1.0f;
		bRec.sampledComponent = 0;
		bRec.sampledType = EGlossyReflection;
		Vector H = normalize(bRec.wo + bRec.wi);
		return fresnelConductorExact(dot(bRec.wi, H), m_eta, m_k) * m_specularReflectance->eval(bRec.its) * w.x;
	}
	Spectrum sample(BSDFSamplingRecord &bRec, const Point2 &sample) const {
		Float pdf_;
		return dj_beckmann_conductor::sample(bRec, pdf_, sample);
	}

	void serialize(Stream *stream, InstanceManager *manager) const { BSDF::serialize(stream, manager); }
	Float getRoughness(const Intersection &its, int component) const { return 0.5f * (m_alpha1 + m_alpha2); }
	std::string toString() const { return "dj_beckmannconductor[engine = libdjb_hip (MI355X)]"; }
	MTS_DECLARE_CLASS()
private:
	djb::beckmann *m_brdf;
	ref<Texture> m_specularReflectance, m_leanmap1, m_leanmap2;
	Spectrum m_eta, m_k;
	Float m_alpha1, m_alpha2, m_alphaAngle, m_scale;
};

MTS_IMPLEMENT_CLASS_S(dj_beckmann_conductor, false, BSDF)
MTS_EXPORT_PLUGIN(dj_beckmann_conductor, "dj_beckmannconductor BRDF (MI355X engine)")
MTS_NAMESPACE_END
