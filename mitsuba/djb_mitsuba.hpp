// mitsuba/djb_mitsuba.hpp -- what the five Mitsuba 0.5 BSDF shells of this directory share.
//
// The shells are drop-ins for jdupuy/dj_brdf's mitsuba/{dj_merl,dj_utia,dj_abc,dj_sgd,dj_beckmannconductor}.cpp:
// same plugin names, same scene properties and children, same component flags, same guards (including the
// reference's quirks, each cited where it is kept), same values handed back to the integrator.  What differs is
// where the djb:: arithmetic runs: behind include/djb_hip.hpp, i.e. in libdjb_hip.so (MI355X kernels; single-hit
// calls are answered by the library's host twin of the object).  tests/test_mitsuba_shells.py compiles
// these files AND the reference's against the same functional stand-in of the Mitsuba API and requires equal
// outputs, flags and queried property names.
//
// NOT COMPILED INTO A RENDERER IN THIS REPOSITORY (the Mitsuba SDK is absent from the image): copy the directory into
// mitsuba/src/bsdfs/, add  plugins += env.SharedLibrary('dj_merl', ['dj_merl.cpp'], LIBS=['djb_hip'], CPPPATH=[...])
// per shell to src/bsdfs/SConscript; INTEGRATION.md has the wavefront (batched) form, which is the one that uses the GPU.
#pragma once
#include <mitsuba/core/fresolver.h>
#include <mitsuba/render/bsdf.h>
#include <mitsuba/render/texture.h>
#include <mitsuba/hw/basicshader.h>
#include <mitsuba/core/warp.h>

#include "djb_hip.hpp"

MTS_NAMESPACE_BEGIN
namespace djb_mts {

// Mitsuba traces from the eye: its wi is djb's outgoing (viewer) direction o, its wo djb's incident (light) direction i
// (dj_brdf.h:23-26; every reference shell swaps them this way, e.g. mitsuba/dj_merl.cpp:62-63)
inline djb::vec3 dir(const Vector &v) { return djb::vec3(v.x, v.y, v.z); }
inline Spectrum rgb(const djb::vec3 &c) { return Color3(c.x, c.y, c.z); }

inline bool lobe_masked(const BSDFSamplingRecord &bRec, unsigned int lobe) { return !(bRec.typeMask & lobe); }
inline bool other_component(const BSDFSamplingRecord &bRec) { return bRec.component != -1 && bRec.component != 0; }
inline bool at_or_below(const Vector &v) { return Frame::cosTheta(v) <= 0; }

// "reflectance" wins over "diffuseReflectance"; default .5 (mitsuba/dj_merl.cpp:23-25 and the same lines of the other shells).
// The texture only feeds the VPL preview shader, but the properties must be consumed: Mitsuba rejects unqueried ones.
inline Texture *reflectance_property(const Properties &props)
{
	return new ConstantSpectrumTexture(props.getSpectrum(
		props.hasProperty("reflectance") ? "reflectance" : "diffuseReflectance", Spectrum(.5f)));
}
inline bool is_reflectance_child(const std::string &name, ConfigurableObject *child)
{
	return child->getClass()->derivesFrom(MTS_CLASS(Texture)) && (name == "reflectance" || name == "diffuseReflectance");
}
inline fs::path resolved(const std::string &name) { return Thread::getThread()->getFileResolver()->resolve(name); }

// what every glossy shell writes into the record next to the sampled direction
inline void glossy_sample(BSDFSamplingRecord &bRec, const Vector &wo)
{
	bRec.wo = wo;
	bRec.eta = 1.0f;
	bRec.sampledComponent = 0;
	bRec.sampledType = BSDF::EGlossyReflection;
}
// direction sampled on a fitted lobe, value = measured evalp / lobe pdf: the tail shared by dj_merl / dj_abc / dj_sgd
// (mitsuba/dj_merl.cpp:83-97, dj_abc.cpp:87-103).  `pdf_of` is the shell's own BSDF::pdf, guards included, as there.
template <class Shell>
inline Spectrum finish_lobe_sample(const Shell &shell, const djb::brdf &measured, BSDFSamplingRecord &bRec,
                                   const djb::vec3 &i, const djb::vec3 &o)
{
	glossy_sample(bRec, Vector(i.x, i.y, i.z));
	if (at_or_below(bRec.wo))
		return Spectrum(0.0f);
	return rgb(measured.evalp(i, o) / shell.pdf(bRec, ESolidAngle));
}

inline std::string id_only(const char *plugin, const std::string &id)
{
	std::ostringstream oss;
	oss << plugin << "[" << endl << "  id = \"" << id << "\"," << endl << "]";
	return oss.str();
}

} // namespace djb_mts

// The VPL-preview shader of the four measured-material shells: a diffuse stand-in driven by the reflectance texture
// (same GLSL as mitsuba/dj_merl.cpp:147-181; renderer UI, no djb math).  One class per plugin, as Mitsuba's class
// registry wants.
#define DJB_MTS_PREVIEW_SHADER(shader_name) \
class shader_name : public Shader { \
public: \
	shader_name(Renderer *renderer, const Texture *reflectance) : Shader(renderer, EBSDFShader), m_reflectance(reflectance) \
	{ m_reflectanceShader = renderer->registerShaderForResource(m_reflectance.get()); } \
	bool isComplete() const { return m_reflectanceShader.get() != NULL; } \
	void cleanup(Renderer *renderer) { renderer->unregisterShaderForResource(m_reflectance.get()); } \
	void putDependencies(std::vector<Shader *> &deps) { deps.push_back(m_reflectanceShader.get()); } \
	void generateCode(std::ostringstream &oss, const std::string &evalName, const std::vector<std::string> &depNames) const { \
		oss << "vec3 " << evalName << "(vec2 uv, vec3 wi, vec3 wo) {" << endl \
			<< "    if (cosTheta(wi) < 0.0 || cosTheta(wo) < 0.0)" << endl \
			<< "    	return vec3(0.0);" << endl \
			<< "    return " << depNames[0] << "(uv) * inv_pi * cosTheta(wo);" << endl \
			<< "}" << endl << endl \
			<< "vec3 " << evalName << "_diffuse(vec2 uv, vec3 wi, vec3 wo) {" << endl \
			<< "    return " << evalName << "(uv, wi, wo);" << endl \
			<< "}" << endl; \
	} \
	MTS_DECLARE_CLASS() \
private: \
	ref<const Texture> m_reflectance; \
	ref<Shader> m_reflectanceShader; \
};

MTS_NAMESPACE_END
