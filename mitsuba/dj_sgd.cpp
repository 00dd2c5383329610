// mitsuba/dj_sgd.cpp -- Mitsuba 0.5 BSDF plugin "dj_sgd" on top of the MI355X engine (drop-in for jdupuy/dj_brdf
// mitsuba/dj_sgd.cpp:19-191).  The BSDF itself is mitsuba/djb_mitsuba_model.hpp; specific to dj_sgd: the reflectance
// texture is a real member -- a texture child replaces it (l.112-119), it is serialized / unserialized with the instance
// manager (l.35, 122-126) and printed by toString (l.128-135).
#include "djb_mitsuba_model.hpp"

MTS_NAMESPACE_BEGIN
using namespace djb_mts;

class dj_sgd : public model_shell<djb::sgd> {
public:
	dj_sgd(const Properties &props) : model_shell<djb::sgd>(props) {}
	dj_sgd(Stream *stream, InstanceManager *manager) : model_shell<djb::sgd>(stream, manager) {
		m_reflectance = static_cast<Texture *>(manager->getInstance(stream));
		configure();
	}

	void addChild(const std::string &name, ConfigurableObject *child) {
		if (is_reflectance_child(name, child))
			m_reflectance = static_cast<Texture *>(child);
		else
			BSDF::addChild(name, child);
	}
	void serialize(Stream *stream, InstanceManager *manager) const {
		BSDF::serialize(stream, manager);
		manager->serialize(stream, m_reflectance.get());
	}
	std::string toString() const {
		std::ostringstream oss;
		oss << "dj_sgd[" << endl
			<< "  id = \"" << getID() << "\"," << endl
			<< "  reflectance = " << indent(m_reflectance->toString()) << endl
			<< "]";
		return oss.str();
	}
	Shader *createShader(Renderer *renderer) const;
	MTS_DECLARE_CLASS()
};

DJB_MTS_PREVIEW_SHADER(dj_sgd_shader)
Shader *dj_sgd::createShader(Renderer *renderer) const { return new dj_sgd_shader(renderer, m_reflectance.get()); }

MTS_IMPLEMENT_CLASS(dj_sgd_shader, false, Shader)
MTS_IMPLEMENT_CLASS_S(dj_sgd, false, BSDF)
MTS_EXPORT_PLUGIN(dj_sgd, "dj_sgd BRDF")
MTS_NAMESPACE_END
