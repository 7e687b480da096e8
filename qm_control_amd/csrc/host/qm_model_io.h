// qm_model_io.h — host-side ingestion of robot.urdf / task.info / reference.info (see qm_model_io.cpp)
#pragma once
#include <string>
#include <vector>
#include "../../../include/qmhip_layout.h"
namespace qmio {
bool fileExists(const std::string& path);
bool loadEeFrameName(const std::string& taskInfo, std::string& name, std::string& err);
bool buildModelBlob(const std::string& urdf, const std::string& referenceInfo, const std::string& eeFrame, double* mb, std::vector<std::string>* jointNames, std::string& err);
bool buildSettingsBlob(const std::string& taskInfo, const double* mb, double* st, std::string& err);
bool validateModelBlob(const double* mb, std::string& err);
}
