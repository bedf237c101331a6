/* jni_min.h -- NOT the JDK's jni.h.  A minimal stand-in that declares, with the JNI specification's signatures, exactly the
 * pieces bmq_jni.c uses, so that the binding can be compile-checked in an image that has no JDK (tests/test_host.py).  A real
 * build includes $JAVA_HOME/include/jni.h instead (cc -DBMQ_REAL_JNI -I$JAVA_HOME/include -I$JAVA_HOME/include/linux ...). */
#ifndef BMQ_JNI_MIN_H
#define BMQ_JNI_MIN_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef jint jsize;
struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jbyteArray;
typedef jarray jlongArray;
typedef jarray jintArray;
struct _jmethodID;
typedef struct _jmethodID* jmethodID;
#define JNI_OK 0
#define JNI_VERSION_1_8 0x00010008
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNIInvokeInterface_;
typedef const struct JNIInvokeInterface_* JavaVM;
struct JNINativeInterface_ { /* only the members used by bmq_jni.c; the real table has 230 entries */
    jclass (*FindClass)(JNIEnv* env, const char* name);
    jint (*ThrowNew)(JNIEnv* env, jclass clazz, const char* msg);
    void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
    jlong (*GetDirectBufferCapacity)(JNIEnv* env, jobject buf);
    jsize (*GetArrayLength)(JNIEnv* env, jarray array);
    jbyte* (*GetByteArrayElements)(JNIEnv* env, jbyteArray array, jboolean* isCopy);
    void (*ReleaseByteArrayElements)(JNIEnv* env, jbyteArray array, jbyte* elems, jint mode);
    void (*SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
    jobject (*NewDirectByteBuffer)(JNIEnv* env, void* address, jlong capacity);
    /* for callbacks from native threads (routeCacheGetAsync) */
    jobject (*NewGlobalRef)(JNIEnv* env, jobject obj);
    void (*DeleteGlobalRef)(JNIEnv* env, jobject ref);
    jclass (*GetObjectClass)(JNIEnv* env, jobject obj);
    jmethodID (*GetMethodID)(JNIEnv* env, jclass clazz, const char* name, const char* sig);
    void (*CallVoidMethod)(JNIEnv* env, jobject obj, jmethodID method, ...);
    jintArray (*NewIntArray)(JNIEnv* env, jsize len);
    void (*SetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, const jint* buf);
    void (*DeleteLocalRef)(JNIEnv* env, jobject ref);
    jboolean (*ExceptionCheck)(JNIEnv* env);
    void (*ExceptionClear)(JNIEnv* env);
    jint (*GetJavaVM)(JNIEnv* env, JavaVM** vm);
    jbyteArray (*NewByteArray)(JNIEnv* env, jsize len);
    void (*SetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, const jbyte* buf);
};
struct JNIInvokeInterface_ { /* only the members used */
    jint (*GetEnv)(JavaVM* vm, void** penv, jint version);
    jint (*AttachCurrentThreadAsDaemon)(JavaVM* vm, void** penv, void* args);
    jint (*DetachCurrentThread)(JavaVM* vm);
};
#endif
