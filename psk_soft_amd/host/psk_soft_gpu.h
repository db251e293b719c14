// psk_soft_gpu.h -- the C++ host side of the drop-in: a psk_soft_i-shaped component class whose
// serviceFunction() keeps the reference's packet handling (reference cpp/psk_soft.cpp:349-363,
// 400-404, 605-617) and hands the loop to libpsk_soft_hip.so through the C ABI.
//
// The class is a template over the BULKIO port types so that the same source is used
//   * inside REDHAWK:  psk_soft_gpu::component<bulkio::InFloatPort, bulkio::OutFloatPort,
//                                             bulkio::OutShortPort>   (see INTEGRATION.md), and
//   * in this repository's tests: with the small in-memory ports of harness.cpp.
// Port surface used (exactly what the reference uses): InPort::dataTransfer with dataBuffer,
// SRI.{xdelta,mode}, sriChanged, inputQueueFlushed, T, EOS, streamID; getPacket(timeout);
// OutPort::pushSRI(SRI) and pushPacket(vector&, T, EOS, streamID).
//
// Member and method names follow the reference (cpp/psk_soft_base.h:45-68, cpp/psk_soft.h:56-87).
#ifndef PSK_SOFT_GPU_H
#define PSK_SOFT_GPU_H

#include <stdexcept>
#include <string>
#include <vector>

#include "psk_soft_hip.h"

namespace psk_soft_gpu {

enum { NOOP = PSK_SOFT_NOOP, NORMAL = PSK_SOFT_NORMAL };
const float BLOCKING = -1.0f;  // bulkio::Const::BLOCKING

template <class InFloatPort, class OutFloatPort, class OutShortPort>
class component {
  public:
    // properties (cpp/psk_soft_base.h:45-56, defaults cpp/psk_soft_base.cpp:96-148)
    unsigned short samplesPerBaud;
    unsigned int numAvg;
    unsigned short constelationSize;
    unsigned short phaseAvg;
    bool differentialDecoding;
    bool resetState;
    // ports (cpp/psk_soft_base.h:58-68); owned by the caller / the generated base class
    InFloatPort *dataFloat_in;
    OutFloatPort *softDecision_dataFloat_out;
    OutShortPort *bits_dataShort_out;
    OutFloatPort *phase_dataFloat_out;
    OutShortPort *sampleIndex_dataShort_out;
    int warnings;  // LOG_WARN count (cpp/psk_soft.cpp:355,361,566)

    // device: HIP device ordinal, or PSK_SOFT_DEVICE_NONE for a control-plane-only component
    component(int device, unsigned max_packet_complex = 1u << 20, unsigned max_window_samples = 65536,
              unsigned max_phase_avg = 4096)
        : samplesPerBaud(10), numAvg(100), constelationSize(4), phaseAvg(50), differentialDecoding(false),
          resetState(false), dataFloat_in(0), softDecision_dataFloat_out(0), bits_dataShort_out(0),
          phase_dataFloat_out(0), sampleIndex_dataShort_out(0), warnings(0), handle_(0)
    {
        psk_soft_limits_t lim;
        lim.max_window_samples = max_window_samples;
        lim.max_phase_avg = max_phase_avg;
        lim.max_packet_complex = max_packet_complex;
        psk_soft_status st = psk_soft_create(device, 1, &lim, &handle_);
        if (st != PSK_SOFT_OK)
            throw std::runtime_error(std::string("psk_soft_create: ") + psk_soft_last_error());
    }
    ~component() { psk_soft_destroy(handle_); }

    // the three registered listeners (cpp/psk_soft.cpp:210-212, 638-651): forward the new values at
    // once, the library decides which reset flag the change sets
    void samplesPerBaudChanged(const std::string &) { push_properties(); }
    void constelationSizeChanged(const std::string &) { push_properties(); }
    void phaseAvgChanged(const std::string &) { push_properties(); }

    int serviceFunction()
    {
        typename InFloatPort::dataTransfer *tmp = dataFloat_in->getPacket(BLOCKING);
        if (!tmp)  // cpp/psk_soft.cpp:350-352
            return NOOP;
        push_properties();  // numAvg / differentialDecoding / resetState have no listener (:374-378)

        psk_soft_packet_t pkt;
        pkt.data = tmp->dataBuffer.empty() ? 0 : &tmp->dataBuffer[0];
        pkt.n_floats = tmp->dataBuffer.size();
        pkt.sri_xdelta = tmp->SRI.xdelta;
        pkt.sri_mode = tmp->SRI.mode;
        pkt.sriChanged = tmp->sriChanged ? 1 : 0;
        pkt.inputQueueFlushed = tmp->inputQueueFlushed ? 1 : 0;
        pkt.present = 1;
        pkt.reserved = 0;

        const size_t cap = (size_t)psk_soft_output_capacity(handle_, 0, pkt.n_floats / 2);
        out_.assign(2 * cap, 0.0f);
        bits_.assign(3 * cap, 0);
        phase_vec_.assign(cap, 0.0f);
        sampleIndexOut_.assign(cap, 0);
        psk_soft_output_t o;
        o.soft = out_.empty() ? 0 : &out_[0];
        o.bits = bits_.empty() ? 0 : &bits_[0];
        o.phase = phase_vec_.empty() ? 0 : &phase_vec_[0];
        o.sampleIndex = sampleIndexOut_.empty() ? 0 : &sampleIndexOut_[0];
        o.cap_symbols = cap;
        psk_soft_status st = psk_soft_process_host(handle_, 0, 1, &pkt, &o);
        if (st != PSK_SOFT_OK) {
            std::string msg = std::string("psk_soft_process_host: ") + psk_soft_last_error();
            delete tmp;
            throw std::runtime_error(msg);
        }
        warnings += o.n_warn;
        resetState = false;  // consumed by the library (:365-372); inputQueueFlushed sets and clears it there
        psk_soft_props_t now;
        if (psk_soft_query(handle_, 0, &now) == PSK_SOFT_OK)
            pushed_ = now;

        if (o.sri_pushed) {  // cpp/psk_soft.cpp:399-404
            tmp->SRI.xdelta = o.sri_soft_xdelta;
            softDecision_dataFloat_out->pushSRI(tmp->SRI);
            tmp->SRI.mode = 0;
            phase_dataFloat_out->pushSRI(tmp->SRI);
            tmp->SRI.xdelta = o.sri_bits_xdelta;
            bits_dataShort_out->pushSRI(tmp->SRI);
        }
        out_.resize(2 * o.n_symbols);
        bits_.resize(o.n_bits);
        phase_vec_.resize(o.n_symbols);
        sampleIndexOut_.resize(o.n_sampleIndex);
        if (!out_.empty())  // cpp/psk_soft.cpp:605-615
            softDecision_dataFloat_out->pushPacket(out_, tmp->T, tmp->EOS, tmp->streamID);
        if (!bits_.empty())
            bits_dataShort_out->pushPacket(bits_, tmp->T, tmp->EOS, tmp->streamID);
        if (!phase_vec_.empty())
            phase_dataFloat_out->pushPacket(phase_vec_, tmp->T, tmp->EOS, tmp->streamID);
        if (!sampleIndexOut_.empty())
            sampleIndex_dataShort_out->pushPacket(sampleIndexOut_, tmp->T, tmp->EOS, tmp->streamID);
        delete tmp;  // the reference leaks the packet on its real-data path (:359-363); we do not
        return NORMAL;
    }

    psk_soft_handle_t *handle() { return handle_; }

  private:
    void push_properties()
    {
        psk_soft_props_t p;
        p.samplesPerBaud = samplesPerBaud;
        p.constelationSize = constelationSize;
        p.numAvg = numAvg;
        p.phaseAvg = phaseAvg;
        p.differentialDecoding = differentialDecoding ? 1 : 0;
        p.resetState = resetState ? 1 : 0;
        psk_soft_status st = psk_soft_configure(handle_, 0, 1, &p);
        if (st != PSK_SOFT_OK)
            throw std::runtime_error(std::string("psk_soft_configure: ") + psk_soft_last_error());
        pushed_ = p;
    }

    psk_soft_handle_t *handle_;
    psk_soft_props_t pushed_;
    std::vector<float> out_;
    std::vector<short> bits_;
    std::vector<float> phase_vec_;
    std::vector<short> sampleIndexOut_;
};

}  // namespace psk_soft_gpu
#endif
