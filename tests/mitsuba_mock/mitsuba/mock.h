// tests/mitsuba_mock -- a MINIMAL stand-in for the part of the Mitsuba 0.5 plugin API that the five shells under
// mitsuba/ use.  TEST INFRASTRUCTURE ONLY (tests/test_mitsuba_shells_syntax.py runs `g++ -fsyntax-only` over the
// shells with this on the include path): declarations written from the call sites of the shells themselves, no
// Mitsuba source, no behaviour.  It lets a typo in a shell fail the CPU suite; it proves nothing about Mitsuba.
#pragma once
#include <cmath>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#define MTS_NAMESPACE_BEGIN namespace mitsuba {
#define MTS_NAMESPACE_END }
#define MTS_DECLARE_CLASS() virtual const Class *getClass() const; static Class *m_theClass;
#define MTS_IMPLEMENT_CLASS(name, abstract, super) Class *name::m_theClass = 0; const Class *name::getClass() const { return m_theClass; }
#define MTS_IMPLEMENT_CLASS_S(name, abstract, super) MTS_IMPLEMENT_CLASS(name, abstract, super)
#define MTS_EXPORT_PLUGIN(name, descr) extern "C" void *CreateInstance(const Properties &props) { return new name(props); } \
	extern "C" const char *GetDescription() { return descr; }
#define MTS_CLASS(x) x::m_theClass

namespace boost { std::string to_lower_copy(const std::string &s); }
namespace fs { struct path { path(); path(const std::string &); std::string string() const; path filename() const; }; }

namespace mitsuba {
typedef float Float;
struct Class { bool derivesFrom(const Class *) const; };
struct Vector { Float x, y, z; Vector(); Vector(Float, Float, Float); };
Vector operator+(const Vector &, const Vector &);
Vector normalize(const Vector &);
Float dot(const Vector &, const Vector &);
struct Point2 { Float x, y; };
struct Frame { static Float cosTheta(const Vector &); };
Float degToRad(Float);

template <class T> struct ref {
	ref(); ref(T *); T *operator->() const; operator T *() const; ref &operator=(T *);
};
struct Stream; struct InstanceManager; struct Renderer;
struct Object { virtual ~Object(); virtual const Class *getClass() const; static Class *m_theClass; virtual std::string toString() const; };
struct ConfigurableObject : Object { virtual void addChild(const std::string &, ConfigurableObject *); virtual void configure(); static Class *m_theClass; };

struct ContinuousSpectrum {};
struct InterpolatedSpectrum : ContinuousSpectrum { InterpolatedSpectrum(const fs::path &); };
struct Spectrum {
	Spectrum(); explicit Spectrum(Float);
	void fromLinearRGB(Float, Float, Float); void toLinearRGB(Float &, Float &, Float &) const;
	void fromContinuousSpectrum(const ContinuousSpectrum &);
	Spectrum operator/(Float) const; Spectrum operator*(Float) const; Spectrum operator*(const Spectrum &) const;
};
Spectrum fresnelConductorExact(Float cosThetaI, const Spectrum &eta, const Spectrum &k);

struct Properties {
	std::string getString(const std::string &) const; std::string getString(const std::string &, const std::string &) const;
	Float getFloat(const std::string &, Float) const; Spectrum getSpectrum(const std::string &, const Spectrum &) const;
	bool hasProperty(const std::string &) const;
};
Float lookupIOR(const Properties &, const std::string &, const std::string &);
struct FileResolver : Object { fs::path resolve(const std::string &) const; };
struct Thread { static Thread *getThread(); FileResolver *getFileResolver(); };

struct Intersection {};
struct Texture : ConfigurableObject { virtual Spectrum eval(const Intersection &, bool filter = true) const; static Class *m_theClass; };
struct ConstantSpectrumTexture : Texture { ConstantSpectrumTexture(const Spectrum &); };
struct ConstantFloatTexture : Texture { ConstantFloatTexture(Float); };

enum EMeasure { ESolidAngle = 1, EDiscrete = 2 };
struct BSDFSamplingRecord {
	const Intersection &its; Vector wi, wo; Float eta; unsigned int typeMask, sampledType; int sampledComponent;
};
struct Shader : Object {
	enum EShaderType { EBSDFShader = 0 };
	Shader(Renderer *, EShaderType);
	virtual void generateCode(std::ostringstream &, const std::string &, const std::vector<std::string> &) const;
	static Class *m_theClass;
};
struct BSDF : ConfigurableObject {
	enum EBSDFType { EDiffuseReflection = 0x1, EGlossyReflection = 0x4, EAnisotropic = 0x1000, ESpatiallyVarying = 0x2000,
	                 EFrontSide = 0x10000 };
	BSDF(const Properties &); BSDF(Stream *, InstanceManager *);
	virtual void configure(); virtual void addChild(const std::string &, ConfigurableObject *);
	virtual void serialize(Stream *, InstanceManager *) const;
	virtual Spectrum eval(const BSDFSamplingRecord &, EMeasure) const = 0;
	virtual Float pdf(const BSDFSamplingRecord &, EMeasure) const = 0;
	virtual Spectrum sample(BSDFSamplingRecord &, Float &, const Point2 &) const = 0;
	virtual Spectrum sample(BSDFSamplingRecord &, const Point2 &) const = 0;
	virtual Float getRoughness(const Intersection &, int) const = 0;
	virtual Shader *createShader(Renderer *) const;
	static Class *m_theClass;
protected:
	std::vector<unsigned int> m_components; bool m_usesRayDifferentials;
};
namespace warp { Vector squareToCosineHemisphere(const Point2 &); Float squareToCosineHemispherePdf(const Vector &); }
} // namespace mitsuba
