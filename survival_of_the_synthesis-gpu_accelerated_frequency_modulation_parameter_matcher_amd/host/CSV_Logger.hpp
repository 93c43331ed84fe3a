// CSV_Logger.hpp -- same interface and file format as the reference's CSV_Logger
// (CSV_Logger.hpp:8-56): a header row of field names, then one comma-terminated row per
// record.  Rewritten for this backend; not a copy.
#ifndef SOTS_CSV_LOGGER_HPP
#define SOTS_CSV_LOGGER_HPP

#include <cstddef>
#include <fstream>
#include <string>
#include <vector>

class CSV_Logger
{
public:
    size_t recordLength_;
    std::ofstream csvFile;

    CSV_Logger(const std::string aFilePath, std::vector<std::string> aFields) : recordLength_(aFields.size())
    {
        if (!aFilePath.empty()) csvFile.open(aFilePath);
        writeRow(aFields);
    }
    bool close()
    {
        if (csvFile.is_open()) csvFile.close();
        return true;
    }
    // A record of the wrong width is rejected, as in the reference (CSV_Logger.hpp:30-31).
    bool addRecord(std::vector<std::string> aRecord)
    {
        if (aRecord.size() != recordLength_) return false;
        writeRow(aRecord);
        return true;
    }
    bool addField(std::string aField)
    {
        if (csvFile.is_open()) csvFile << aField << ",";
        return true;
    }
    bool endRecord()
    {
        if (csvFile.is_open()) csvFile << "\n";
        return true;
    }

private:
    void writeRow(const std::vector<std::string> &aCells)
    {
        if (!csvFile.is_open()) return;
        for (const std::string &c : aCells) csvFile << c << ",";
        csvFile << "\n";
    }
};

#endif
