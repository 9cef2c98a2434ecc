/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref/libgnsssdr_ref_chan.so) -- never linked into the product.
 *
 * The reference's own adapters by implementation name, the way GNSSBlockFactory::GetAcqBlock / GetTrkBlock dispatch on <role>.implementation
 * (src/core/receiver/gnss_block_factory.cc:449-700): tests/host/test_channel.cc builds the reference's Channel (channel.cc, compiled in place into the same
 * library) over these, beside the same Channel over the HIP adapters.  The test includes only the reference's interface headers; the adapters' own headers
 * (GNU Radio / VOLK / armadillo stand-ins and all) stay inside this library.
 */
#ifndef GSH_ORACLE_REF_CHAN_API_H
#define GSH_ORACLE_REF_CHAN_API_H
#include "acquisition_interface.h"
#include "configuration_interface.h"
#include "tracking_interface.h"
#include <memory>
#include <string>

std::shared_ptr<AcquisitionInterface> refchan_make_acquisition(const std::string& implementation, const ConfigurationInterface* configuration, const std::string& role,
    unsigned int in_streams, unsigned int out_streams);
std::shared_ptr<TrackingInterface> refchan_make_tracking(const std::string& implementation, const ConfigurationInterface* configuration, const std::string& role,
    unsigned int in_streams, unsigned int out_streams);
#endif
