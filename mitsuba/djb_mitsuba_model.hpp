// mitsuba/djb_mitsuba_model.hpp -- the part dj_abc and dj_sgd share (jdupuy/dj_brdf mitsuba/dj_abc.cpp:19-141 and
// mitsuba/dj_sgd.cpp:19-144 are the same BSDF around djb::abc / djb::sgd; they differ in how the reflectance texture is
// serialized, which stays in the two .cpp files).
//
// eval = the analytic model's evalp; sample / pdf = djb::tabular(model, 90), a tabulated lobe fitted at load time on the
// GPU that samples with the non-VNDF "nmap" scheme (supports_smith_vndf_sampling() == false).
// Reference quirks kept:
//   * the material is read from "merlID" and the NAME goes through the file resolver (dj_abc.cpp:29-30, dj_sgd.cpp:28-29);
//     the resolver returns unknown names unchanged, and the whole resolved string is what djb::abc / djb::sgd receive;
//   * the component is registered as EGlossyReflection (l.51) but eval tests EDiffuseReflection (l.56) -- so an
//     integrator that asks for glossy lobes only gets pdf / sample from this BSDF and a black eval;
//   * pdf honours bRec.component, eval does not; sample rejects cosTheta(wi) < 0, not <= 0 (l.83).
#pragma once
#include "djb_mitsuba.hpp"

MTS_NAMESPACE_BEGIN
namespace djb_mts {

template <class Model>
class model_shell : public BSDF {
public:
	model_shell(const Properties &props) : BSDF(props), m_model(NULL), m_tabular(NULL) {
		m_reflectance = reflectance_property(props);
		const std::string merlID = resolved(props.getString("merlID")).string();
		m_model = new Model(merlID.c_str());              // djb::exc "No ... parameters for <name>" on an unknown material
		m_tabular = new djb::tabular(*m_model, 90);
	}
	model_shell(Stream *stream, InstanceManager *manager) : BSDF(stream, manager), m_model(NULL), m_tabular(NULL) {}
	~model_shell() { delete m_tabular; delete m_model; }

	void configure() {
		m_components.clear();
		m_components.push_back(EGlossyReflection | EFrontSide);
		m_usesRayDifferentials = false;
		BSDF::configure();
	}

	Spectrum eval(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (lobe_masked(bRec, EDiffuseReflection) || measure != ESolidAngle || at_or_below(bRec.wi) || at_or_below(bRec.wo))
			return Spectrum(0.0f);
		return rgb(m_model->evalp(dir(bRec.wo), dir(bRec.wi)));
	}
	Float pdf(const BSDFSamplingRecord &bRec, EMeasure measure) const {
		if (measure != ESolidAngle || at_or_below(bRec.wi) || at_or_below(bRec.wo)
			|| other_component(bRec) || lobe_masked(bRec, EGlossyReflection))
			return 0.0f;
		return m_tabular->pdf(dir(bRec.wo), dir(bRec.wi));
	}
	Spectrum sample(BSDFSamplingRecord &bRec, const Point2 &sample) const {
		if (Frame::cosTheta(bRec.wi) < 0 || other_component(bRec) || lobe_masked(bRec, EGlossyReflection))
			return Spectrum(0.0f);
		const djb::vec3 o = dir(bRec.wi);
		return finish_lobe_sample(*this, *m_model, bRec, m_tabular->sample(sample.x, sample.y, o), o);
	}
	Spectrum sample(BSDFSamplingRecord &bRec, Float &pdf_, const Point2 &sample_) const {
		Spectrum res = sample(bRec, sample_);
		pdf_ = pdf(bRec, ESolidAngle);
		return res;
	}
	void serialize(Stream *stream, InstanceManager *manager) const { BSDF::serialize(stream, manager); }
protected:
	ref<const Texture> m_reflectance;
	Model *m_model;
	djb::tabular *m_tabular;
};

} // namespace djb_mts
MTS_NAMESPACE_END
