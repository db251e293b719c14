// harness.cpp -- in-memory ports + a C API so that Python tests can drive the C++ host class
// (psk_soft_gpu.h) the way the reference's test drives the sandbox-launched component
// (reference tests/test_psk_soft.py:119-269): configure properties, push a packet into the
// input port, run serviceFunction(), read what the four output ports received.
// Test support only; not part of libpsk_soft_hip.so.
#include <deque>
#include <string>
#include <vector>

#include "psk_soft_gpu.h"

namespace {

struct StreamSRI {
    double xdelta;
    int mode;
    std::string streamID;
};
struct Time {
    double twsec, tfsec;
};

struct InFloatPort {
    struct dataTransfer {
        std::vector<float> dataBuffer;
        StreamSRI SRI;
        Time T;
        bool EOS;
        std::string streamID;
        bool sriChanged;
        bool inputQueueFlushed;
    };
    std::deque<dataTransfer *> q;
    dataTransfer *getPacket(float)
    {
        if (q.empty())
            return 0;
        dataTransfer *p = q.front();
        q.pop_front();
        return p;
    }
    ~InFloatPort()
    {
        for (size_t i = 0; i < q.size(); i++) delete q[i];
    }
};

template <class T>
struct OutPort {
    std::vector<T> data;           // payload of every pushPacket since the last drain
    std::vector<double> sri_xdelta;  // xdelta of every pushSRI
    std::vector<int> sri_mode;
    int n_packets;
    bool last_eos;
    std::string last_stream;
    double last_twsec;
    OutPort() : n_packets(0), last_eos(false), last_twsec(0) {}
    void pushSRI(const StreamSRI &s)
    {
        sri_xdelta.push_back(s.xdelta);
        sri_mode.push_back(s.mode);
    }
    void pushPacket(std::vector<T> &v, const Time &t, bool eos, const std::string &id)
    {
        data.insert(data.end(), v.begin(), v.end());
        n_packets++;
        last_eos = eos;
        last_stream = id;
        last_twsec = t.twsec;
    }
};

typedef psk_soft_gpu::component<InFloatPort, OutPort<float>, OutPort<short> > Component;

struct Harness {
    InFloatPort in;
    OutPort<float> soft, phase;
    OutPort<short> bits, sidx;
    Component comp;
    std::string error;
    explicit Harness(int device) : comp(device)
    {
        comp.dataFloat_in = &in;
        comp.softDecision_dataFloat_out = &soft;
        comp.bits_dataShort_out = &bits;
        comp.phase_dataFloat_out = &phase;
        comp.sampleIndex_dataShort_out = &sidx;
    }
};

}  // namespace

extern "C" {

void *psk_harness_create(int device, char *err, int errlen)
{
    try {
        return new Harness(device);
    } catch (const std::exception &e) {
        if (err && errlen > 0) {
            std::string m = e.what();
            size_t n = m.size() < (size_t)errlen - 1 ? m.size() : (size_t)errlen - 1;
            for (size_t i = 0; i < n; i++) err[i] = m[i];
            err[n] = 0;
        }
        return 0;
    }
}
void psk_harness_destroy(void *h) { delete (Harness *)h; }

// configure() of one property: stores it and, like REDHAWK's PropertySet, runs the registered
// change listener when the value changed.  id: 0 samplesPerBaud 1 numAvg 2 constelationSize
// 3 phaseAvg 4 differentialDecoding 5 resetState
int psk_harness_configure(void *hv, int id, unsigned value)
{
    Harness *h = (Harness *)hv;
    Component &c = h->comp;
    try {
        switch (id) {
        case 0: { bool ch = c.samplesPerBaud != value; c.samplesPerBaud = (unsigned short)value; if (ch) c.samplesPerBaudChanged("samplesPerBaud"); break; }
        case 1: c.numAvg = value; break;
        case 2: { bool ch = c.constelationSize != value; c.constelationSize = (unsigned short)value; if (ch) c.constelationSizeChanged("constelationSize"); break; }
        case 3: { bool ch = c.phaseAvg != value; c.phaseAvg = (unsigned short)value; if (ch) c.phaseAvgChanged("phaseAvg"); break; }
        case 4: c.differentialDecoding = value != 0; break;
        case 5: c.resetState = value != 0; break;
        default: return -1;
        }
    } catch (const std::exception &e) {
        h->error = e.what();
        return -2;
    }
    return 0;
}

void psk_harness_push(void *hv, const float *data, size_t n_floats, double xdelta, int mode, int sriChanged,
                      int flushed, int eos, const char *streamID, double twsec)
{
    Harness *h = (Harness *)hv;
    InFloatPort::dataTransfer *p = new InFloatPort::dataTransfer();
    p->dataBuffer.assign(data, data + n_floats);
    p->SRI.xdelta = xdelta;
    p->SRI.mode = mode;
    p->SRI.streamID = streamID ? streamID : "";
    p->T.twsec = twsec;
    p->T.tfsec = 0;
    p->EOS = eos != 0;
    p->streamID = p->SRI.streamID;
    p->sriChanged = sriChanged != 0;
    p->inputQueueFlushed = flushed != 0;
    h->in.q.push_back(p);
}

int psk_harness_service(void *hv)
{
    Harness *h = (Harness *)hv;
    try {
        return h->comp.serviceFunction();
    } catch (const std::exception &e) {
        h->error = e.what();
        return -1;
    }
}
const char *psk_harness_error(void *hv) { return ((Harness *)hv)->error.c_str(); }
int psk_harness_warnings(void *hv) { return ((Harness *)hv)->comp.warnings; }

// port: 0 soft 1 bits 2 phase 3 sampleIndex
static size_t port_size(Harness *h, int port)
{
    switch (port) {
    case 0: return h->soft.data.size();
    case 1: return h->bits.data.size();
    case 2: return h->phase.data.size();
    default: return h->sidx.data.size();
    }
}
size_t psk_harness_port_size(void *hv, int port) { return port_size((Harness *)hv, port); }
int psk_harness_port_packets(void *hv, int port)
{
    Harness *h = (Harness *)hv;
    switch (port) {
    case 0: return h->soft.n_packets;
    case 1: return h->bits.n_packets;
    case 2: return h->phase.n_packets;
    default: return h->sidx.n_packets;
    }
}
// copies and drains the accumulated payload of one port (getData() of a sandbox DataSink)
size_t psk_harness_port_read_f32(void *hv, int port, float *dst, size_t cap)
{
    Harness *h = (Harness *)hv;
    std::vector<float> &v = port == 0 ? h->soft.data : h->phase.data;
    size_t n = v.size() < cap ? v.size() : cap;
    for (size_t i = 0; i < n; i++) dst[i] = v[i];
    v.clear();
    return n;
}
size_t psk_harness_port_read_i16(void *hv, int port, short *dst, size_t cap)
{
    Harness *h = (Harness *)hv;
    std::vector<short> &v = port == 1 ? h->bits.data : h->sidx.data;
    size_t n = v.size() < cap ? v.size() : cap;
    for (size_t i = 0; i < n; i++) dst[i] = v[i];
    v.clear();
    return n;
}
// pushSRI log of one port: returns count, fills xdelta / mode of the most recent `cap` entries
size_t psk_harness_port_sri(void *hv, int port, double *xdelta, int *mode, size_t cap)
{
    Harness *h = (Harness *)hv;
    std::vector<double> *x;
    std::vector<int> *m;
    switch (port) {
    case 0: x = &h->soft.sri_xdelta; m = &h->soft.sri_mode; break;
    case 1: x = &h->bits.sri_xdelta; m = &h->bits.sri_mode; break;
    case 2: x = &h->phase.sri_xdelta; m = &h->phase.sri_mode; break;
    default: x = &h->sidx.sri_xdelta; m = &h->sidx.sri_mode; break;
    }
    size_t n = x->size();
    for (size_t i = 0; i < n && i < cap; i++) {
        xdelta[i] = (*x)[i];
        mode[i] = (*m)[i];
    }
    return n;
}
int psk_harness_last_eos(void *hv) { return ((Harness *)hv)->soft.last_eos ? 1 : 0; }
const char *psk_harness_last_stream(void *hv) { return ((Harness *)hv)->soft.last_stream.c_str(); }
void *psk_harness_handle(void *hv) { return ((Harness *)hv)->comp.handle(); }

}  // extern "C"
