// tests/mitsuba_mock -- a FUNCTIONAL minimal stand-in for the part of the Mitsuba 0.5 plugin API that the Mitsuba BSDF shells
// of dj_brdf use.  TEST INFRASTRUCTURE ONLY.
//
// Purpose: pin the GLUE of this repository's shells (mitsuba/*.cpp) to the glue of the reference's shells.  The same
// header-only mock is put under (a) the reference's own five shells, compiled unchanged from /root/reference/mitsuba in the
// build container against /root/reference/dj_brdf.h, and (b) this repository's shells compiled against include/djb_hip.hpp +
// libdjb_hip.so.  tests/mitsuba_mock/shell_harness.cpp drives both through the plugin entry point (CreateInstance) and
// BSDF::eval / pdf / sample; the reference side's outputs are committed as tests/golden/shells.npz and the repository side
// must reproduce them (tests/test_mitsuba_shells.py).  Every property name a constructor queries is recorded, so a shell
// that reads "material" where the reference reads "merlID" fails the test.
//
// What it is NOT: Mitsuba.  Spectrum is 3 linear-RGB floats; warp / fresnelConductorExact / lookupIOR / InterpolatedSpectrum
// are deterministic stand-ins written from the textbook formulas (or, for the .spd loader, a hash of the file name) --
// good enough because both sides run the SAME stand-in.  A green test says "same calls, same order, same arguments
// into Mitsuba-side code", nothing about Mitsuba's own arithmetic ("parity unpinned" for those factors).
// Written from the call sites of the shells; contains no Mitsuba source.
//
// Include order: <cmath> only, never <math.h> -- the canonical oracle order of SURVEY.md 8-N.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define MTS_NAMESPACE_BEGIN namespace mitsuba {
#define MTS_NAMESPACE_END }
#define MTS_CLASS(x) (&x::m_theClass)
#define MTS_DECLARE_CLASS() virtual const Class *getClass() const; static Class m_theClass;
#define MTS_IMPLEMENT_CLASS(name, abstract, super) Class name::m_theClass(#name, &super::m_theClass); \
	const Class *name::getClass() const { return &m_theClass; }
#define MTS_IMPLEMENT_CLASS_S(name, abstract, super) MTS_IMPLEMENT_CLASS(name, abstract, super)
#define MTS_EXPORT_PLUGIN(name, descr) extern "C" { \
	void *CreateInstance(const mitsuba::Properties *props) { return new mitsuba::name(*props); } \
	void *CreateInstanceFromStream(mitsuba::Stream *s, mitsuba::InstanceManager *m) { return new mitsuba::name(s, m); } \
	const char *GetDescription() { return descr; } }

namespace boost {
inline std::string to_lower_copy(const std::string &s)
{
	std::string r(s);
	for (size_t k = 0; k < r.size(); ++k) if (r[k] >= 'A' && r[k] <= 'Z') r[k] = (char)(r[k] - 'A' + 'a');
	return r;
}
}
namespace fs {
struct path {
	path() {}
	path(const std::string &s) : m_s(s) {}
	path(const char *s) : m_s(s) {}
	std::string string() const { return m_s; }
	path filename() const { size_t p = m_s.rfind('/'); return p == std::string::npos ? *this : path(m_s.substr(p + 1)); }
	std::string m_s;
};
}

namespace mitsuba {
using std::endl;
typedef float Float;

inline std::string indent(const std::string &s)
{
	std::string r;
	for (size_t k = 0; k < s.size(); ++k) { r += s[k]; if (s[k] == '\n') r += "  "; }
	return r;
}

enum ELogLevel { ETrace = 0, EDebug = 100, EInfo = 200, EWarn = 300, EError = 400 };
inline void SLog(ELogLevel level, const char *fmt, ...)
{
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	if (level >= EError) throw std::runtime_error(buf);      // Mitsuba's EError raises
	fprintf(stderr, "%s\n", buf);
}

// ---- class registry (getClass()->derivesFrom(MTS_CLASS(Texture)))
struct Class {
	Class(const char *name, const Class *super) : m_name(name), m_super(super) {}
	bool derivesFrom(const Class *c) const { for (const Class *k = this; k; k = k->m_super) if (k == c) return true; return false; }
	const char *m_name; const Class *m_super;
};
struct Object {
	virtual ~Object() {}
	virtual const Class *getClass() const { return &m_theClass; }
	virtual std::string toString() const { return "Object[]"; }
	static Class m_theClass;
};
// one definition per shared object is enough for the test harness (each shell is a single translation unit)
#ifdef MITSUBA_MOCK_MAIN
Class Object::m_theClass("Object", NULL);
#endif

// ---- ref<T>: a plain owning-nothing handle (the test leaks; it never shares objects across owners)
template <class T> struct ref {
	ref() : m_p(NULL) {}
	ref(T *p) : m_p(p) {}
	template <class U> ref(const ref<U> &o) : m_p(o.get()) {}
	ref &operator=(T *p) { m_p = p; return *this; }
	T *operator->() const { return m_p; }
	T *get() const { return m_p; }
	operator T *() const { return m_p; }
	T *m_p;
};
template <class A, class B> inline bool operator!=(const ref<A> &a, const ref<B> &b) { return a.get() != b.get(); }
template <class A, class B> inline bool operator==(const ref<A> &a, const ref<B> &b) { return a.get() == b.get(); }

// ---- geometry
struct Vector {
	Float x, y, z;
	Vector() : x(0), y(0), z(0) {}
	Vector(Float x_, Float y_, Float z_) : x(x_), y(y_), z(z_) {}
};
struct Normal : Vector { Normal() {} Normal(Float x_, Float y_, Float z_) : Vector(x_, y_, z_) {} };
inline Vector operator+(const Vector &a, const Vector &b) { return Vector(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Float dot(const Vector &a, const Vector &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vector normalize(const Vector &v) { Float r = 1.0f / std::sqrt(dot(v, v)); return Vector(v.x * r, v.y * r, v.z * r); }
struct Point2 { Float x, y; Point2() : x(0), y(0) {} Point2(Float x_, Float y_) : x(x_), y(y_) {} };
struct Frame { static Float cosTheta(const Vector &v) { return v.z; } };
inline Float degToRad(Float d) { return d * (Float)(M_PI / 180.0); }

// ---- serialization: an in-memory byte stream; the instance manager hands objects back in the order they were stored
struct Stream {
	std::vector<unsigned char> bytes; size_t pos;
	Stream() : pos(0) {}
	void write(const void *p, size_t n) { const unsigned char *c = (const unsigned char *)p; bytes.insert(bytes.end(), c, c + n); }
	void read(void *p, size_t n) { if (pos + n > bytes.size()) throw std::runtime_error("Stream: read past the end"); memcpy(p, &bytes[pos], n); pos += n; }
	void writeFloat(Float f) { write(&f, sizeof f); }
	Float readFloat() { Float f; read(&f, sizeof f); return f; }
	void writeString(const std::string &s) { unsigned int n = (unsigned int)s.size(); write(&n, sizeof n); write(s.data(), n); }
	std::string readString() { unsigned int n; read(&n, sizeof n); std::string s(n, ' '); if (n) read(&s[0], n); return s; }
};
struct ConfigurableObject;
struct InstanceManager {
	std::vector<const ConfigurableObject *> stored; size_t next;
	InstanceManager() : next(0) {}
	void serialize(Stream *stream, const ConfigurableObject *obj) { stream->writeString("<instance>"); stored.push_back(obj); }
	ConfigurableObject *getInstance(Stream *stream)
	{
		if (stream->readString() != "<instance>" || next >= stored.size()) throw std::runtime_error("InstanceManager: stream out of step");
		return const_cast<ConfigurableObject *>(stored[next++]);
	}
};

// ---- spectra: 3 linear-RGB floats
struct ContinuousSpectrum { virtual ~ContinuousSpectrum() {} virtual Float probe(int c) const = 0; };
struct Spectrum {
	Float s[3];
	Spectrum() { s[0] = s[1] = s[2] = 0; }
	explicit Spectrum(Float v) { s[0] = s[1] = s[2] = v; }
	Spectrum(Float r, Float g, Float b) { s[0] = r; s[1] = g; s[2] = b; }
	explicit Spectrum(Stream *stream) { for (int c = 0; c < 3; ++c) s[c] = stream->readFloat(); }
	void serialize(Stream *stream) const { for (int c = 0; c < 3; ++c) stream->writeFloat(s[c]); }
	void fromLinearRGB(Float r, Float g, Float b) { s[0] = r; s[1] = g; s[2] = b; }
	void toLinearRGB(Float &r, Float &g, Float &b) const { r = s[0]; g = s[1]; b = s[2]; }
	void fromContinuousSpectrum(const ContinuousSpectrum &c) { for (int k = 0; k < 3; ++k) s[k] = c.probe(k); }
	Float average() const { return (s[0] + s[1] + s[2]) * (1.0f / 3.0f); }
	Spectrum operator/(Float f) const { Float r = 1.0f / f; return Spectrum(s[0] * r, s[1] * r, s[2] * r); }
	Spectrum operator*(Float f) const { return Spectrum(s[0] * f, s[1] * f, s[2] * f); }
	Spectrum operator*(const Spectrum &o) const { return Spectrum(s[0] * o.s[0], s[1] * o.s[1], s[2] * o.s[2]); }
	std::string toString() const { char b[96]; snprintf(b, sizeof b, "[%.9g, %.9g, %.9g]", s[0], s[1], s[2]); return b; }
};
inline Spectrum operator*(Float f, const Spectrum &v) { return v * f; }
typedef Spectrum Color3;
// stand-in for the .spd loader: three values in (0.1, 3.3) derived from the file name (FNV-1a), so that a shell which
// resolves a different file gets different optical constants
struct InterpolatedSpectrum : ContinuousSpectrum {
	explicit InterpolatedSpectrum(const fs::path &p) : m_path(p.string()) {}
	Float probe(int c) const
	{
		unsigned int h = 2166136261u;
		for (size_t k = 0; k < m_path.size(); ++k) h = (h ^ (unsigned char)m_path[k]) * 16777619u;
		h = (h ^ (unsigned int)(c + 1)) * 16777619u;
		return 0.1f + (Float)((h >> 8) & 0xFFFF) * (3.2f / 65535.0f);
	}
	std::string m_path;
};
// textbook exact Fresnel reflectance of a conductor (unpolarised), per channel
inline Spectrum fresnelConductorExact(Float cosThetaI, const Spectrum &eta, const Spectrum &k)
{
	Spectrum r;
	Float c2 = cosThetaI * cosThetaI, s2 = 1 - c2, s4 = s2 * s2;
	for (int c = 0; c < 3; ++c) {
		Float t1 = eta.s[c] * eta.s[c] - k.s[c] * k.s[c] - s2;
		Float a2pb2 = std::sqrt(t1 * t1 + 4 * k.s[c] * k.s[c] * eta.s[c] * eta.s[c]);
		Float a = std::sqrt(0.5f * (a2pb2 + t1));
		Float term1 = a2pb2 + c2, term2 = 2 * a * cosThetaI;
		Float Rs2 = (term1 - term2) / (term1 + term2);
		Float term3 = a2pb2 * c2 + s4, term4 = term2 * s2;
		Float Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
		r.s[c] = 0.5f * (Rp2 + Rs2);
	}
	return r;
}

// ---- scene description: a typed property map that records every name asked for
struct Properties {
	enum EType { EString, EFloat, EBoolean, ESpectrum };
	struct Value { EType type; std::string str; Float f; bool b; Spectrum spec; };
	std::map<std::string, Value> values;
	mutable std::vector<std::string> queried;                 // in query order, duplicates kept
	std::string id;
	void note(const std::string &n) const { queried.push_back(n); }
	const Value *find(const std::string &n, EType t) const
	{
		std::map<std::string, Value>::const_iterator it = values.find(n);
		if (it == values.end()) return NULL;
		if (it->second.type != t) throw std::runtime_error("Property \"" + n + "\" has a different type");
		return &it->second;
	}
	bool hasProperty(const std::string &n) const { note(n); return values.count(n) != 0; }
	std::string getString(const std::string &n) const
	{ note(n); const Value *v = find(n, EString); if (!v) throw std::runtime_error("Property \"" + n + "\" missing"); return v->str; }
	std::string getString(const std::string &n, const std::string &d) const { note(n); const Value *v = find(n, EString); return v ? v->str : d; }
	Float getFloat(const std::string &n, Float d) const { note(n); const Value *v = find(n, EFloat); return v ? v->f : d; }
	bool getBoolean(const std::string &n, bool d) const { note(n); const Value *v = find(n, EBoolean); return v ? v->b : d; }
	Spectrum getSpectrum(const std::string &n, const Spectrum &d) const { note(n); const Value *v = find(n, ESpectrum); return v ? v->spec : d; }
	std::string getID() const { return id; }
};
// src/bsdfs/ior.h: a float "extEta" wins, else the named dielectric ("air")
inline Float lookupIOR(const Properties &props, const std::string &name, const std::string &def)
{
	if (props.hasProperty(name)) return props.getFloat(name, 1.0f);
	return def == "air" ? 1.000277f : 1.5046f;
}
struct FileResolver : Object { fs::path resolve(const fs::path &p) const { resolved().push_back(p.string()); return p; }
	static std::vector<std::string> &resolved() { static std::vector<std::string> r; return r; } };
struct Thread {
	static Thread *getThread() { static Thread t; return &t; }
	FileResolver *getFileResolver() { static FileResolver f; return &f; }
};

// ---- textures
struct Intersection { Float u, v; Intersection() : u(0), v(0) {} };
struct ConfigurableObject : Object {
	virtual void addChild(const std::string &name, ConfigurableObject *) { throw std::runtime_error("addChild: unsupported child \"" + name + "\""); }
	virtual void configure() {}
	virtual const Class *getClass() const { return &m_theClass; }
	static Class m_theClass;
};
struct Texture : ConfigurableObject {
	virtual Spectrum eval(const Intersection &its, bool filter = true) const = 0;
	virtual bool isConstant() const { return false; }
	virtual bool usesRayDifferentials() const { return false; }
	virtual const Class *getClass() const { return &m_theClass; }
	static Class m_theClass;
};
struct ConstantSpectrumTexture : Texture {
	explicit ConstantSpectrumTexture(const Spectrum &v) : m_v(v) {}
	Spectrum eval(const Intersection &, bool = true) const { return m_v; }
	bool isConstant() const { return true; }
	std::string toString() const { return "ConstantSpectrumTexture[value=" + m_v.toString() + "]"; }
	Spectrum m_v;
};
struct ConstantFloatTexture : Texture {
	explicit ConstantFloatTexture(Float v) : m_v(v) {}
	Spectrum eval(const Intersection &, bool = true) const { return Spectrum(m_v); }
	bool isConstant() const { return true; }
	std::string toString() const { char b[64]; snprintf(b, sizeof b, "ConstantFloatTexture[value=%.9g]", m_v); return b; }
	Float m_v;
};
// a spatially varying test texture: value = a + b * u + c * v per channel (the harness attaches it as a child)
struct AffineTexture : Texture {
	AffineTexture(const Spectrum &a, const Spectrum &bu, const Spectrum &cv) : m_a(a), m_bu(bu), m_cv(cv) {}
	Spectrum eval(const Intersection &its, bool = true) const
	{ return Spectrum(m_a.s[0] + m_bu.s[0] * its.u + m_cv.s[0] * its.v, m_a.s[1] + m_bu.s[1] * its.u + m_cv.s[1] * its.v,
	                  m_a.s[2] + m_bu.s[2] * its.u + m_cv.s[2] * its.v); }
	std::string toString() const { return "AffineTexture[" + m_a.toString() + "]"; }
	Spectrum m_a, m_bu, m_cv;
};
#ifdef MITSUBA_MOCK_MAIN
Class ConfigurableObject::m_theClass("ConfigurableObject", &Object::m_theClass);
Class Texture::m_theClass("Texture", &ConfigurableObject::m_theClass);
#endif

// ---- hardware preview shaders (VPL renderer): enough of the interface for the shells' shader classes to be driven
struct Renderer;
struct GPUProgram {
	mutable std::vector<std::string> names; mutable std::vector<Spectrum> values;
	int getParameterID(const std::string &name, bool = true) const { names.push_back(name); values.push_back(Spectrum()); return (int)names.size() - 1; }
	void setParameter(int id, const Spectrum &v) { values.at((size_t)id) = v; }
};
struct Shader : Object {
	enum EShaderType { EBSDFShader = 0, ETextureShader = 1 };
	Shader(Renderer *, EShaderType t) : m_type(t) {}
	virtual bool isComplete() const { return true; }
	virtual void putDependencies(std::vector<Shader *> &) {}
	virtual void cleanup(Renderer *) {}
	virtual void resolve(const GPUProgram *, const std::string &, std::vector<int> &) const {}
	virtual void bind(GPUProgram *, const std::vector<int> &, int &) const {}
	virtual void generateCode(std::ostringstream &, const std::string &, const std::vector<std::string> &) const {}
	virtual const Class *getClass() const { return &m_theClass; }
	static Class m_theClass;
	EShaderType m_type;
};
struct Renderer {
	std::map<const void *, Shader *> shaders; int registered, unregistered;
	Renderer() : registered(0), unregistered(0) {}
	Shader *registerShaderForResource(const Texture *t)
	{
		++registered;
		if (!shaders.count(t)) shaders[t] = new Shader(this, Shader::ETextureShader);
		return shaders[t];
	}
	void unregisterShaderForResource(const Texture *) { ++unregistered; }
};
#ifdef MITSUBA_MOCK_MAIN
Class Shader::m_theClass("Shader", &Object::m_theClass);
#endif

// ---- BSDF interface
enum EMeasure { EInvalidMeasure = 0, ESolidAngle = 1, ELength = 2, EArea = 3, EDiscrete = 4 };
struct BSDFSamplingRecord {
	explicit BSDFSamplingRecord(const Intersection &its_) : its(its_), eta(0), typeMask(0xFFFFFFFFu), sampledType(0), component(-1), sampledComponent(-1) {}
	const Intersection &its; Vector wi, wo; Float eta; unsigned int typeMask, sampledType; int component, sampledComponent;
};
struct BSDF : ConfigurableObject {
	enum EBSDFType { ENull = 0x1, EDiffuseReflection = 0x2, EDiffuseTransmission = 0x4, EGlossyReflection = 0x8,
	                 EGlossyTransmission = 0x10, EDeltaReflection = 0x20, EDeltaTransmission = 0x40,
	                 EAnisotropic = 0x1000, ESpatiallyVarying = 0x2000, ENonSymmetric = 0x4000,
	                 EFrontSide = 0x8000, EBackSide = 0x10000, EUsesSampler = 0x20000 };
	explicit BSDF(const Properties &props) : m_usesRayDifferentials(false), m_combinedType(0), m_id(props.getID()), m_configured(0) {}
	BSDF(Stream *stream, InstanceManager *) : m_usesRayDifferentials(false), m_combinedType(0), m_id(stream->readString()), m_configured(0) {}
	virtual void configure() { m_combinedType = 0; for (size_t k = 0; k < m_components.size(); ++k) m_combinedType |= m_components[k]; ++m_configured; }
	virtual void addChild(const std::string &name, ConfigurableObject *child) { ConfigurableObject::addChild(name, child); }
	virtual void serialize(Stream *stream, InstanceManager *) const { stream->writeString(m_id); }
	virtual Spectrum eval(const BSDFSamplingRecord &, EMeasure) const = 0;
	virtual Float pdf(const BSDFSamplingRecord &, EMeasure) const = 0;
	virtual Spectrum sample(BSDFSamplingRecord &, Float &, const Point2 &) const = 0;
	virtual Spectrum sample(BSDFSamplingRecord &, const Point2 &) const = 0;
	virtual Float getRoughness(const Intersection &, int) const { return -1.0f; }       // "not overridden" marker
	virtual Shader *createShader(Renderer *) const { return NULL; }
	virtual const Class *getClass() const { return &m_theClass; }
	const std::string &getID() const { return m_id; }
	// (texture, name, max) -> the texture; Mitsuba wraps it in a scaling texture when it can exceed max
	Texture *ensureEnergyConservation(Texture *t, const std::string &name, Float) const { m_ensured.push_back(name); return t; }
	static Class m_theClass;
	// harness access
	std::vector<unsigned int> m_components; bool m_usesRayDifferentials; unsigned int m_combinedType;
	std::string m_id; int m_configured; mutable std::vector<std::string> m_ensured;
};
#ifdef MITSUBA_MOCK_MAIN
Class BSDF::m_theClass("BSDF", &ConfigurableObject::m_theClass);
#endif

// ---- warps (concentric disk -> cosine hemisphere)
namespace warp {
inline Vector squareToCosineHemisphere(const Point2 &sample)
{
	Float r1 = 2.0f * sample.x - 1.0f, r2 = 2.0f * sample.y - 1.0f, phi, r;
	if (r1 == 0 && r2 == 0) { r = phi = 0; }
	else if (r1 * r1 > r2 * r2) { r = r1; phi = (Float)(M_PI / 4.0) * (r2 / r1); }
	else { r = r2; phi = (Float)(M_PI / 2.0) - (r1 / r2) * (Float)(M_PI / 4.0); }
	Float x = r * std::cos(phi), y = r * std::sin(phi);
	Float z = std::sqrt(std::max((Float)1e-10f, 1.0f - x * x - y * y));
	return Vector(x, y, z);
}
inline Float squareToCosineHemispherePdf(const Vector &d) { return (Float)(1.0 / M_PI) * Frame::cosTheta(d); }
}
} // namespace mitsuba
