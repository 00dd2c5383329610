// mitsuba/dj_abc.cpp -- Mitsuba 0.5 BSDF plugin "dj_abc" on top of the MI355X engine (drop-in for jdupuy/dj_brdf
// mitsuba/dj_abc.cpp:19-191).  The BSDF itself is mitsuba/djb_mitsuba_model.hpp; this file holds what is specific to
// dj_abc: the reflectance child is accepted and dropped (l.113-119) and nothing but the base record is serialized (l.122-125).
#include "djb_mitsuba_model.hpp"

MTS_NAMESPACE_BEGIN
using namespace djb_mts;

class dj_abc : public model_shell<djb::abc> {
public:
	dj_abc(const Properties &props) : model_shell<djb::abc>(props) {}
	dj_abc(Stream *stream, InstanceManager *manager) : model_shell<djb::abc>(stream, manager) { configure(); }

	void addChild(const std::string &name, ConfigurableObject *child) {
		if (!is_reflectance_child(name, child))
			BSDF::addChild(name, child);
	}
	std::string toString() const { return id_only("dj_abc", getID()); }
	Shader *createShader(Renderer *renderer) const;
	MTS_DECLARE_CLASS()
};

DJB_MTS_PREVIEW_SHADER(dj_abc_shader)
Shader *dj_abc::createShader(Renderer *renderer) const { return new dj_abc_shader(renderer, m_reflectance.get()); }

MTS_IMPLEMENT_CLASS(dj_abc_shader, false, Shader)
MTS_IMPLEMENT_CLASS_S(dj_abc, false, BSDF)
MTS_EXPORT_PLUGIN(dj_abc, "MERL BRDF")
MTS_NAMESPACE_END
