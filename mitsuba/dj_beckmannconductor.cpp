// mitsuba/dj_beckmannconductor.cpp -- Mitsuba 0.5 BSDF plugin "dj_beckmannconductor" (class dj_beckmann_conductor) on top of
// the MI355X engine: drop-in for jdupuy/dj_brdf mitsuba/dj_beckmannconductor.cpp:151-582 (see mitsuba/djb_mitsuba.hpp).
//
// A Beckmann conductor whose lobe is rebuilt per hit from LEAN / LEADR maps.  Per hit the reference does (l.291-314, again at
// 339-362 and 379-402):
//     params = elliptic(alpha1(its), alpha2(its), alphaAngle(its))            alpha* are TEXTURES evaluated at the hit
//     E1..E5 = texels of leanmap1 (E1, E2) and leanmap2 (E3, E4, E5);  E1 -= 25, E2 -= 25, E5 -= 625      (always, l.300-303)
//     lrep1  = leanFiltering ? lrep(E1, E2, E3, E4, E5) : lrep(E1, E2, E1*E1, E2*E2, E1*E2)
//     lrep1 *= dmapscale;  params = lrep_to_params(lrep1 + params_to_lrep(params))
// and then evalp / pdf / evalp_is with those params.  Here that whole block plus the operator is ONE library call per hit:
// beckmann::evalp_lean / evalp_is_lean (djb_eval_lean_batch / djb_sample_lean_batch with n = 1 and
// DJB_LEAN_BIASED [| DJB_LEAN_NAIVE_MIP]); a wavefront renderer passes n hits to the same calls.
// The exact conductor Fresnel factor stays Mitsuba's (fresnelConductorExact: its arithmetic lives in the renderer).
//
// Reference behaviour kept, quirks included:
//   * material defaults to "none" (ideal mirror) unless mitsubaFresnel is set, then "Cu" (l.160-162);
//   * with a "merl" property the Beckmann lobe carries the MERL-fitted Fresnel spline, beckmann(tab->get_fresnel()), and every
//     alpha is MULTIPLIED by the fitted base roughness (l.179-213); the default alpha is the base roughness itself (1 without merl);
//   * eval and pdf do not reject directions below the horizon (the cosTheta tests are commented out, l.282-283, 333-334);
//     sample does not either (l.374);
//   * the +25 / +625 texel bias is subtracted even when no LEAN map is given (constant-0 default textures, l.216-221);
//   * ESpatiallyVarying looks at alpha1 / alpha2 / specularReflectance only, not at the LEAN maps (l.259-261);
//   * the unserializing constructor restores the six textures and eta / k but none of the scalar options.
#include "djb_mitsuba.hpp"
#include "microfacet.h"
#include "ior.h"

MTS_NAMESPACE_BEGIN
using namespace djb_mts;

class dj_beckmann_conductor : public BSDF {
	// the six texture slots, in the order the reference serializes and prints them (l.224-229, 244-249, 436-441)
	enum slot { ALPHA1, ALPHA2, ALPHA_ANGLE, LEANMAP1, LEANMAP2, SPECULAR, SLOTS };
	static const char *slot_name(int k) {
		static const char *const names[SLOTS] = { "alpha1", "alpha2", "alphaAngle", "leanmap1", "leanmap2", "specularReflectance" };
		return names[k];
	}
public:
	dj_beckmann_conductor(const Properties &props) : BSDF(props), m_brdf(NULL) {
		ref<FileResolver> fResolver = Thread::getThread()->getFileResolver();
		m_tex[SPECULAR] = new ConstantSpectrumTexture(props.getSpectrum(slot_name(SPECULAR), Spectrum(1.0f)));

		// Fresnel: optical constants for Mitsuba's exact conductor term
		m_mitsubaFresnel = props.getBoolean("mitsubaFresnel", false);
		const std::string materialName = props.getString("material", m_mitsubaFresnel ? "Cu" : "none");
		Spectrum intEta(0.0f), intK(1.0f);
		if (boost::to_lower_copy(materialName) != "none") {
			intEta.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + materialName + ".eta.spd")));
			intK.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + materialName + ".k.spd")));
		}
		const Float extEta = lookupIOR(props, "extEta", "air");
		m_eta = props.getSpectrum("eta", intEta) / extEta;
		m_k = props.getSpectrum("k", intK) / extEta;

		// the lobe; with "merl": base roughness and Fresnel spline fitted from the measured material (one fit launch)
		float base = 1.f;
		if (props.hasProperty("merl")) {
			const std::string file = fResolver->resolve(props.getString("merl")).string();
			djb::merl merl(file.c_str());
			djb::tabular tab(merl, 90);
			djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&base, &base);
			m_brdf = new djb::beckmann(tab.get_fresnel());
		} else {
			m_brdf = new djb::beckmann();
		}

		// roughness, each value scaled by the base roughness: "alpha" alone, or "alpha1" together with "alpha2" (l.193-213)
		const bool both = props.hasProperty("alpha");
		const bool split = !both && (props.hasProperty("alpha1") || props.hasProperty("alpha2"));
		if (both) {
			m_tex[ALPHA1] = m_tex[ALPHA2] = new ConstantFloatTexture(base * props.getFloat("alpha", 1.0f));
			if (props.hasProperty("alpha1") || props.hasProperty("alpha2") || props.hasProperty("alphaAngle"))
				SLog(EError, "Microfacet model: please specify either 'alpha' or 'alpha1'/'alpha2'/'alphaAngle'.");
		} else if (split) {
			if (!props.hasProperty("alpha1") || !props.hasProperty("alpha2"))
				SLog(EError, "Microfacet model: both 'alpha1' and 'alpha2' must be specified.");
			for (int k = ALPHA1; k <= ALPHA2; ++k)
				m_tex[k] = new ConstantFloatTexture(base * props.getFloat(slot_name(k), 1.0f));
		} else {
			m_tex[ALPHA1] = m_tex[ALPHA2] = new ConstantFloatTexture(base);
		}
		m_tex[ALPHA_ANGLE] = new ConstantFloatTexture(M_PI / 180.0 * props.getFloat(slot_name(ALPHA_ANGLE), 0.0f));

		// LEAN maps
		m_leanFiltering = props.getBoolean("leanFiltering", true);
		for (int k = LEANMAP1; k <= LEANMAP2; ++k)
			m_tex[k] = new ConstantSpectrumTexture(props.getSpectrum(slot_name(k), Spectrum(0.0f)));
		m_dmapScale = props.getFloat("dmapscale", 1.0f);
	}

	// restores the textures and eta / k, none of the scalar options (as the reference, l.223-234)
	dj_beckmann_conductor(Stream *stream, InstanceManager *manager) : BSDF(stream, manager), m_brdf(NULL) {
		for (int k = 0; k < SLOTS; ++k)
			m_tex[k] = static_cast<Texture *>(manager->getInstance(stream));
		m_eta = Spectrum(stream);
		m_k = Spectrum(stream);
		configure();
	}
	~dj_beckmann_conductor() { delete m_brdf; }

	void serialize(Stream *stream, InstanceManager *manager) const {
		BSDF::serialize(stream, manager);
		for (int k = 0; k < SLOTS; ++k)
			manager->serialize(stream, m_tex[k].get());
		m_eta.serialize(stream);
		m_k.serialize(stream);
	}

	void configure() {
		// the LEAN maps and the angle count for neither flag (l.256-268)
		const Texture *seen[3] = { m_tex[ALPHA1].get(), m_tex[ALPHA2].get(), m_tex[SPECULAR].get() };
		bool varying = false, differentials = false;
		unsigned int flags = EGlossyReflection | EFrontSide;
		if (seen[0] != seen[1])
			flags |= EAnisotropic;
		for (int k = 0; k < 3; ++k)
			varying |= !seen[k]->isConstant();
		if (varying)
			flags |= ESpatiallyVarying;
		m_components.clear();
		m_components.push_back(flags);
		m_tex[SPECULAR] = ensureEnergyConservation(m_tex[SPECULAR], slot_name(SPECULAR), 1.0f);
		seen[2] = m_tex[SPECULAR].get();
		for (int k = 0; k < 3; ++k)
			differentials |= seen[k]->usesRayDifferentials();
		m_usesRayDifferentials = differentials;
		BSDF::configure();
	}

	Spectrum eval(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (measure != ESolidAngle || other_component(bRec) || lobe_masked(bRec, EGlossyReflection))
			return Spectrum(0.0f);
		hit h(*this, bRec.its);
		const djb::vec3 o = dir(bRec.wi), i = dir(bRec.wo);
		djb::vec3 fr_cos;
		m_brdf->evalp_lean(1, &i, &o, h.base, m_dmapScale, h.texel, &fr_cos, NULL, h.flags);
		return conductor(bRec) * rgb(fr_cos);
	}

	Float pdf(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (measure != ESolidAngle || other_component(bRec) || lobe_masked(bRec, EGlossyReflection))
			return 0.0f;
		hit h(*this, bRec.its);
		const djb::vec3 o = dir(bRec.wi), i = dir(bRec.wo);
		djb::vec3 fr_cos;
		float density;
		m_brdf->evalp_lean(1, &i, &o, h.base, m_dmapScale, h.texel, &fr_cos, &density, h.flags);
		return density;
	}

	Spectrum sample(BSDFSamplingRecord &bRec, Float &pdf, const Point2 &sample) const {
		if (other_component(bRec) || lobe_masked(bRec, EGlossyReflection))
			return Spectrum(0.0f);
		hit h(*this, bRec.its);
		const djb::vec3 o = dir(bRec.wi);
		djb::vec3 i, fr_cos;
		m_brdf->evalp_is_lean(1, &sample.x, &sample.y, &o, h.base, m_dmapScale, h.texel, &fr_cos, &i, &pdf, h.flags);
		glossy_sample(bRec, Normal(i.x, i.y, i.z));
		return conductor(bRec) * rgb(fr_cos);
	}
	Spectrum sample(BSDFSamplingRecord &bRec, const Point2 &sample) const {
		Float unused = 0.f;
		return this->sample(bRec, unused, sample);
	}

	void addChild(const std::string &name, ConfigurableObject *child) {
		if (child->getClass()->derivesFrom(MTS_CLASS(Texture))) {
			Texture *texture = static_cast<Texture *>(child);
			if (name == "alpha") {
				m_tex[ALPHA1] = m_tex[ALPHA2] = texture;
				return;
			}
			for (int k = 0; k < SLOTS; ++k)
				if (name == slot_name(k)) {
					m_tex[k] = texture;
					return;
				}
		}
		BSDF::addChild(name, child);
	}

	Float getRoughness(const Intersection &its, int component) const {
		return 0.5f * (m_tex[ALPHA1]->eval(its).average() + m_tex[ALPHA2]->eval(its).average());
	}

	std::string toString() const {
		std::ostringstream oss;
		oss << "dj_beckmann_conductor[" << endl
			<< "  id = \"" << getID() << "\"," << endl;
		for (int k = 0; k < SLOTS; ++k)
			oss << "  " << slot_name(k) << " = " << indent(m_tex[k]->toString()) << "," << endl;
		oss << "  eta = " << m_eta.toString() << "," << endl
			<< "  k = " << m_k.toString() << endl
			<< "]";
		return oss.str();
	}

	Shader *createShader(Renderer *renderer) const;
	MTS_DECLARE_CLASS()
private:
	// what one intersection contributes: the base lobe from the alpha textures and the raw LEAN texel (bias still on)
	struct hit {
		hit(const dj_beckmann_conductor &s, const Intersection &its)
			: base(djb::microfacet::params::elliptic(s.m_tex[ALPHA1]->eval(its).average(), s.m_tex[ALPHA2]->eval(its).average(),
			                                         s.m_tex[ALPHA_ANGLE]->eval(its).average())),
			  flags(DJB_LEAN_BIASED | (s.m_leanFiltering ? 0 : DJB_LEAN_NAIVE_MIP))
		{
			Float dummy;
			s.m_tex[LEANMAP1]->eval(its).toLinearRGB(texel[0], texel[1], dummy);
			s.m_tex[LEANMAP2]->eval(its).toLinearRGB(texel[2], texel[3], texel[4]);
		}
		djb::microfacet::params base;
		Float texel[5];
		int flags;
	};
	// l.321-324 / 421-424: exact conductor Fresnel at the half vector times the specular reflectance texture
	Spectrum conductor(const BSDFSamplingRecord &bRec) const {
		const Vector half = normalize(bRec.wo + bRec.wi);
		return fresnelConductorExact(dot(bRec.wi, half), m_eta, m_k) * m_tex[SPECULAR]->eval(bRec.its);
	}

	ref<Texture> m_tex[SLOTS];
	djb::beckmann *m_brdf;
	Spectrum m_eta, m_k;
	Float m_dmapScale;
	bool m_leanFiltering, m_mitsubaFresnel;
};

// GLSL preview (VPL renderer): the reference ships an Ashikhmin-Shirley stand-in with its own GLSL program here
// (mitsuba/dj_beckmannconductor.cpp:456-520).  Preview shaders are renderer UI with no djb math and are out of scope
// (SURVEY.md 2 #21): like the four measured-material shells this one registers the neutral diffuse preview of
// djb_mitsuba.hpp, driven by the specular reflectance texture.
DJB_MTS_PREVIEW_SHADER(dj_beckmann_conductor_shader)

Shader *dj_beckmann_conductor::createShader(Renderer *renderer) const {
	return new dj_beckmann_conductor_shader(renderer, m_tex[SPECULAR].get());
}

MTS_IMPLEMENT_CLASS(dj_beckmann_conductor_shader, false, Shader)
MTS_IMPLEMENT_CLASS_S(dj_beckmann_conductor, false, BSDF)
MTS_EXPORT_PLUGIN(dj_beckmann_conductor, "Rough conductor BRDF");
MTS_NAMESPACE_END
