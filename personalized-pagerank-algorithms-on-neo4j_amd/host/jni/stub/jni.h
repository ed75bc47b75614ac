/* jni.h — NOT the JDK's header.  A declaration-only stand-in for exactly the JNI calls pprhip_jni.cpp makes, so that
 * `make jni-check` can type-check that file in an image without a JDK (nothing links against it, nothing runs with
 * it; the real binding is built with `make jni JAVA_HOME=...` against the JDK's own <jni.h>).  Signatures follow the
 * Java Native Interface Specification (Oracle, "JNI Functions"). */
#ifndef PPRHIP_JNI_STUB_H
#define PPRHIP_JNI_STUB_H
#include <cstdint>

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;
class _jobject {};
typedef _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;
struct _jfieldID;
typedef _jfieldID* jfieldID;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNIEnv {
  jclass GetObjectClass(jobject);
  jclass FindClass(const char*);
  jfieldID GetFieldID(jclass, const char*, const char*);
  jlong GetLongField(jobject, jfieldID);
  void SetLongField(jobject, jfieldID, jlong);
  jint ThrowNew(jclass, const char*);
  jboolean ExceptionCheck();
  jsize GetArrayLength(jarray);
  jint* GetIntArrayElements(jintArray, jboolean*);
  void ReleaseIntArrayElements(jintArray, jint*, jint);
  void GetIntArrayRegion(jintArray, jsize, jsize, jint*);
  void SetIntArrayRegion(jintArray, jsize, jsize, const jint*);
  void GetLongArrayRegion(jlongArray, jsize, jsize, jlong*);
  void SetDoubleArrayRegion(jdoubleArray, jsize, jsize, const jdouble*);
  jintArray NewIntArray(jsize);
  jdoubleArray NewDoubleArray(jsize);
  const char* GetStringUTFChars(jstring, jboolean*);
  void ReleaseStringUTFChars(jstring, const char*);
};
#endif
